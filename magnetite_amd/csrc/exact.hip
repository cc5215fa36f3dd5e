// Reference-order fp64 arithmetic of the path, one kernel per reference
// function.  Compiled with -ffp-contract=off: the Rust reference never fuses
// a*b+c, and nalgebra's static matrix product sums ascending k starting from
// the k=0 product (gemm -> gemv -> axcpy), so K_e, the assembled K, the RHS
// and the stress scalar reproduce the reference's rounding, not just its
// formulas.  None of these kernels is in the CG loop.
#include <cstring>

#include <hip/hip_runtime.h>

#include "kernels.h"

namespace magk {

static inline int blocks_for(int64_t n, int threads)
{
    int64_t b = (n + threads - 1) / threads;
    return (int)(b < 1 ? 1 : b);
}

// C[m x n] = A[m x k] * B[k x n], row major, ascending-k sums from the first product
template <int M, int K, int N>
__device__ inline void matmul(const double *a, const double *b, double *c)
{
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double s = a[i * K] * b[j];
#pragma unroll
            for (int l = 1; l < K; ++l) s = s + a[i * K + l] * b[l * N + j];
            c[i * N + j] = s;
        }
}

// solver.rs:187-193
__device__ inline double signed_area(double x0, double y0, double x1, double y1, double x2, double y2)
{
    return 0.5 * (x0 * (y1 - y2) + x1 * (y2 - y0) + x2 * (y0 - y1));
}

// solver.rs:204-230
__device__ inline void strain_displacement(double x0, double y0, double x1, double y1, double x2, double y2,
                                           double area, double *B)
{
    const double b1 = y1 - y2, b2 = y2 - y0, b3 = y0 - y1;
    const double g1 = x2 - x1, g2 = x0 - x2, g3 = x1 - x0;
    const double m[18] = {b1, 0., b2, 0., b3, 0., 0., g1, 0., g2, 0., g3, g1, b1, g2, b2, g3, b3};
    const double d = 2.0 * area;
#pragma unroll
    for (int i = 0; i < 18; ++i) B[i] = m[i] / d;
}

// solver.rs:240-250
__device__ inline void stress_strain(double nu, double youngs, double *D)
{
    const double m[9] = {1.0, nu, 0.0, nu, 1.0, 0.0, 0.0, 0.0, (1.0 - nu) / 2.0};
    const double s = youngs / (1.0 - nu * nu);
#pragma unroll
    for (int i = 0; i < 9; ++i) D[i] = m[i] * s;
}

// solver.rs:263-278 + the loop of solver.rs:548-567: one thread per element.
__global__ void __launch_bounds__(256) k_element_stiffness(const double2 *xy, const int32_t *conn, int64_t E,
                                                           double nu, double youngs, double thick, double *ke)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const double2 v0 = xy[conn[3 * e]], v1 = xy[conn[3 * e + 1]], v2 = xy[conn[3 * e + 2]];
    const double area = signed_area(v0.x, v0.y, v1.x, v1.y, v2.x, v2.y);
    double D[9], B[18], Bt[18], BtD[18], K[36];
    stress_strain(nu, youngs, D);
    strain_displacement(v0.x, v0.y, v1.x, v1.y, v2.x, v2.y, area, B);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) Bt[j * 3 + i] = B[i * 6 + j];
    matmul<6, 3, 3>(Bt, D, BtD);
    matmul<6, 3, 6>(BtD, B, K);
    double *out = ke + 36 * e;
#pragma unroll
    for (int i = 0; i < 36; ++i) out[i] = K[i] * area * thick;
}

void element_stiffness(const double *xy, const int32_t *conn, int64_t E, double nu, double youngs, double thick,
                       double *ke, hipStream_t s)
{
    k_element_stiffness<<<blocks_for(E, 256), 256, 0, s>>>((const double2 *)xy, conn, E, nu, youngs, thick, ke);
}

// solver.rs:290-331 without the dense matrix and without atomics: the sorted
// (row node, col node) pair list groups every '+=' that lands on one 2x2
// block; inside a group the stable sort kept ascending element order, which
// is the reference's summation order (0.0 + first contribution is exact).
__global__ void __launch_bounds__(256) k_assemble_gather(const uint64_t *keys, const uint32_t *vals,
                                                         const int32_t *seg_start, int64_t nb, const int32_t *bptr,
                                                         const double *ke, double *kval)
{
    const int64_t blk = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (blk >= nb) return;
    const int32_t s0 = seg_start[blk], s1 = seg_start[blk + 1];
    const int32_t i = (int32_t)(keys[s0] >> 32);
    double k00 = 0.0, k01 = 0.0, k10 = 0.0, k11 = 0.0;
    for (int32_t k = s0; k < s1; ++k) {
        const uint32_t v = vals[k];
        const uint32_t e = v / 9u, ab = v - 9u * e;
        const uint32_t a = ab / 3u, b = ab - 3u * a;
        const double *src = ke + 36 * (int64_t)e + 12 * a + 2 * b; // K_e[2a][2b]
        k00 += src[0];
        k01 += src[1];
        k10 += src[6];
        k11 += src[7];
    }
    const int32_t p = bptr[i], cnt = bptr[i + 1] - p, kpos = (int32_t)blk - p;
    double *r0 = kval + 4 * (int64_t)p + 2 * kpos;
    double *r1 = kval + 4 * (int64_t)p + 2 * cnt + 2 * kpos;
    r0[0] = k00;
    r0[1] = k01;
    r1[0] = k10;
    r1[1] = k11;
}

void assemble_gather(const uint64_t *keys, const uint32_t *vals, const int32_t *seg_start, int64_t nb,
                     const int32_t *bptr, const double *ke, double *kval, hipStream_t s)
{
    k_assemble_gather<<<blocks_for(nb, 256), 256, 0, s>>>(keys, vals, seg_start, nb, bptr, ke, kval);
}

// solver.rs:263-278 + 290-331 fused and still atomic-free: one thread per (row node i, col node j) block of K walks
// node i's incident elements in ascending element order (the reference's '+=' order) and, for every element that also
// holds j, evaluates ONLY the 2x2 block K_e[2a..2a+1][2b..2b+1] -- each entry with exactly the operations
// nalgebra performs for it ((B^T D) B, ascending-k sums from the first product, then * area * thickness), so the
// result is bit-identical to scattering full K_e matrices.  No 288-byte-per-element K_e buffer is written or read.
__device__ inline void ke_block(const double2 v0, const double2 v1, const double2 v2, int a, int b, const double *D,
                                double thick, double &k00, double &k01, double &k10, double &k11)
{
    const double area = signed_area(v0.x, v0.y, v1.x, v1.y, v2.x, v2.y);
    const double d = 2.0 * area;
    const double b0 = v1.y - v2.y, b1 = v2.y - v0.y, b2 = v0.y - v1.y;
    const double g0 = v2.x - v1.x, g1 = v0.x - v2.x, g2 = v1.x - v0.x;
    const double z = 0.0 / d; // the structural zeros of B after `strain_displacement_mat /= 2.0 * area`
    // columns 2c and 2c+1 of B (3 rows): (beta_c, 0, gamma_c)^T / d and (0, gamma_c, beta_c)^T / d
    // (selects, not runtime-indexed arrays: those would live in scratch memory)
    const double ba = (a == 0 ? b0 : (a == 1 ? b1 : b2)) / d, ga = (a == 0 ? g0 : (a == 1 ? g1 : g2)) / d;
    const double bb = (b == 0 ? b0 : (b == 1 ? b1 : b2)) / d, gb = (b == 0 ? g0 : (b == 1 ? g1 : g2)) / d;
    const double Bx_a[3] = {ba, z, ga}, By_a[3] = {z, ga, ba};
    const double Bx_b[3] = {bb, z, gb}, By_b[3] = {z, gb, bb};
    // rows 2a, 2a+1 of B^T D: M[r][m] = sum_k B[k][r] D[k][m]
    double Mx[3], My[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        double s = Bx_a[0] * D[m];
        s = s + Bx_a[1] * D[3 + m];
        s = s + Bx_a[2] * D[6 + m];
        Mx[m] = s;
        double t = By_a[0] * D[m];
        t = t + By_a[1] * D[3 + m];
        t = t + By_a[2] * D[6 + m];
        My[m] = t;
    }
    auto dot3 = [](const double *m, const double *c) {
        double s = m[0] * c[0];
        s = s + m[1] * c[1];
        s = s + m[2] * c[2];
        return s;
    };
    k00 = dot3(Mx, Bx_b) * area * thick;
    k01 = dot3(Mx, By_b) * area * thick;
    k10 = dot3(My, Bx_b) * area * thick;
    k11 = dot3(My, By_b) * area * thick;
}

__global__ void __launch_bounds__(256) k_assemble_rows(const int32_t *brow, const int32_t *bcol, const int32_t *bptr,
                                                       int64_t nb, const int32_t *inc_off, const uint32_t *inc,
                                                       const int32_t *iperm, const int32_t *conn, const double2 *xy,
                                                       double nu, double youngs, double thick, double *kval)
{
    const int64_t blk = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (blk >= nb) return;
    const int32_t i = brow[blk], j = bcol[blk];
    double D[9];
    stress_strain(nu, youngs, D);
    const int32_t g = iperm[i]; // incidence lists are keyed by the Hilbert id
    double k00 = 0.0, k01 = 0.0, k10 = 0.0, k11 = 0.0;
    for (int32_t q = inc_off[g]; q < inc_off[g + 1]; ++q) {
        const uint32_t v = inc[q];
        const uint32_t e = v / 3u;
        const int a = (int)(v - 3u * e);
        const int32_t n0 = conn[3 * (int64_t)e], n1 = conn[3 * (int64_t)e + 1], n2 = conn[3 * (int64_t)e + 2];
        if (n0 != j && n1 != j && n2 != j) continue;
        const double2 v0 = xy[n0], v1 = xy[n1], v2 = xy[n2];
        const int32_t nn[3] = {n0, n1, n2};
#pragma unroll
        for (int b = 0; b < 3; ++b) { // ascending local column, as the loops of solver.rs:304-322
            if (nn[b] != j) continue;
            double c00, c01, c10, c11;
            ke_block(v0, v1, v2, a, b, D, thick, c00, c01, c10, c11);
            k00 += c00;
            k01 += c01;
            k10 += c10;
            k11 += c11;
        }
    }
    const int32_t p = bptr[i], cnt = bptr[i + 1] - p, kpos = (int32_t)blk - p;
    double *r0 = kval + 4 * (int64_t)p + 2 * kpos;
    double *r1 = kval + 4 * (int64_t)p + 2 * cnt + 2 * kpos;
    r0[0] = k00;
    r0[1] = k01;
    r1[0] = k10;
    r1[1] = k11;
}

void assemble_rows(const int32_t *brow, const int32_t *bcol, const int32_t *bptr, int64_t nb, const int32_t *inc_off,
                   const uint32_t *inc, const int32_t *iperm, const int32_t *conn, const double *xy, double nu,
                   double youngs, double thick, double *kval, hipStream_t s)
{
    k_assemble_rows<<<blocks_for(nb, 256), 256, 0, s>>>(brow, bcol, bptr, nb, inc_off, inc, iperm, conn,
                                                        (const double2 *)xy, nu, youngs, thick, kval);
}

// solver.rs:263-278 + 290-331 per ELEMENT TILE (the shape BASELINE.json's north star names), still atomic-free and
// bit-exact.  A tile is kAsmNodes consecutive nodes of the Hilbert order -- a compact patch of the mesh -- together
// with every element incident on them; one workgroup per tile:
//   phase 1  one thread per (node, incident element) entry of the tile's slice of the incidence table (a contiguous,
//            coalesced read of `inc`; connectivity and coordinates gathered once per entry): the element's signed
//            area and the seven quotients of `strain_displacement_mat /= 2.0 * area` (solver.rs:187-230) -- every
//            true fp64 division of K_e -- go to LDS, with the element's node ids;
//   phase 2  one thread per 2x2 block (row node i, col node j) of the tile's K rows, consecutive threads on consecutive
//            blocks of a row: it walks node i's entries IN LDS, in ascending element order (the reference's `+=` order,
//            solver.rs:299-325), and for every element that also holds j forms rows 2a, 2a+1 of B^T D and the block
//            ((B^T D) B)[2a..][2b..] * area * thickness with exactly nalgebra's operations (ascending-k sums from the
//            first product), from the staged quotients: multiplications and additions only.
// Against k_assemble_rows (one thread per block re-deriving area, B and nine divisions per matching element straight
// from global memory: ~9 evaluations per element) an element is now evaluated once per incident node of the tile
// (<= 3), its divisions are out of the block loop, and the block loop reads LDS.  A tile whose entries do not fit
// the LDS image (a hub node of valence in the hundreds) takes the k_assemble_rows path for its blocks, same bits.
constexpr int kAsmNodes = 128;  // nodes per assembly tile
constexpr int kAsmCap = 1024;   // incidence entries staged per tile (8 per node; a gmsh-style mesh has ~6)
constexpr int kAsmThreads = 256;

__global__ void __launch_bounds__(kAsmThreads) k_assemble_tiles(const int32_t *bcol, const int32_t *bptr,
                                                                 const int32_t *inc_off, const uint32_t *inc,
                                                                 const uint32_t *perm, const int32_t *conn,
                                                                 const double2 *xy, int64_t N, double nu, double youngs,
                                                                 double thick, double *kval)
{
    __shared__ double s_val[8][kAsmCap];    // bd0 bd1 bd2 gd0 gd1 gd2 z area, entry-minor: conflict-free by entry
    __shared__ int32_t s_nn[3][kAsmCap];    // the element's node ids (caller numbering); corner a in bits 30-31 of [0]
    __shared__ int32_t s_scan[kAsmNodes + 1]; // exclusive scan of the tile's block counts
    __shared__ int32_t s_eoff[kAsmNodes + 1]; // the nodes' entry ranges, relative to the tile's first entry
    const int tid = threadIdx.x;
    const int64_t g0 = (int64_t)blockIdx.x * kAsmNodes;
    const int nn_tile = (int)(N - g0 < kAsmNodes ? N - g0 : kAsmNodes);
    const int32_t q0 = inc_off[g0];
    const int n_ent = inc_off[g0 + nn_tile] - q0;
    const bool staged = n_ent <= kAsmCap;
    double D[9];
    stress_strain(nu, youngs, D);

    if (tid <= nn_tile) s_eoff[tid] = inc_off[g0 + tid] - q0;
    if (tid < nn_tile) {
        const int64_t i = perm[g0 + tid];
        s_scan[tid + 1] = bptr[i + 1] - bptr[i];
    }
    if (tid == 0) s_scan[0] = 0;
    if (staged)
        for (int q = tid; q < n_ent; q += kAsmThreads) {
            const uint32_t v = inc[q0 + q];
            const uint32_t e = v / 3u;
            const int32_t n0 = conn[3 * (int64_t)e], n1 = conn[3 * (int64_t)e + 1], n2 = conn[3 * (int64_t)e + 2];
            const double2 v0 = xy[n0], v1 = xy[n1], v2 = xy[n2];
            const double area = signed_area(v0.x, v0.y, v1.x, v1.y, v2.x, v2.y);
            const double d = 2.0 * area;
            s_val[0][q] = (v1.y - v2.y) / d;
            s_val[1][q] = (v2.y - v0.y) / d;
            s_val[2][q] = (v0.y - v1.y) / d;
            s_val[3][q] = (v2.x - v1.x) / d;
            s_val[4][q] = (v0.x - v2.x) / d;
            s_val[5][q] = (v1.x - v0.x) / d;
            s_val[6][q] = 0.0 / d; // the structural zeros of B after the division
            s_val[7][q] = area;
            s_nn[0][q] = n0 | (int32_t)((v - 3u * e) << 30);
            s_nn[1][q] = n1;
            s_nn[2][q] = n2;
        }
    __syncthreads();
    if (tid == 0) { // 128 short counts: a serial scan by one lane costs less than the barriers of a parallel one
        int32_t run = 0;
        for (int k = 0; k < nn_tile; ++k) {
            const int32_t c = s_scan[k + 1];
            s_scan[k + 1] = run + c;
            run += c;
        }
    }
    __syncthreads();
    const int nb_tile = s_scan[nn_tile];
    for (int idx = tid; idx < nb_tile; idx += kAsmThreads) {
        int lo = 0, hi = nn_tile; // the node n with s_scan[n] <= idx < s_scan[n + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_scan[mid] <= idx) lo = mid; else hi = mid;
        }
        const int kpos = idx - s_scan[lo];
        const int64_t i = perm[g0 + lo];
        const int32_t p = bptr[i], cnt = s_scan[lo + 1] - s_scan[lo];
        const int32_t j = bcol[p + kpos];
        double k00 = 0.0, k01 = 0.0, k10 = 0.0, k11 = 0.0;
        if (staged) {
            for (int q = s_eoff[lo]; q < s_eoff[lo + 1]; ++q) {
                const int32_t w0 = s_nn[0][q];
                const int32_t nn[3] = {w0 & 0x3fffffff, s_nn[1][q], s_nn[2][q]};
                if (nn[0] != j && nn[1] != j && nn[2] != j) continue;
                const int a = (int)((uint32_t)w0 >> 30);
                const double bd0 = s_val[0][q], bd1 = s_val[1][q], bd2 = s_val[2][q];
                const double gd0 = s_val[3][q], gd1 = s_val[4][q], gd2 = s_val[5][q];
                const double z = s_val[6][q], area = s_val[7][q];
                const double ba = a == 0 ? bd0 : (a == 1 ? bd1 : bd2), ga = a == 0 ? gd0 : (a == 1 ? gd1 : gd2);
                // rows 2a, 2a+1 of B^T D, as ke_block forms them: columns (ba, z, ga) and (z, ga, ba) of B
                double Mx[3], My[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    double sx = ba * D[m];
                    sx = sx + z * D[3 + m];
                    sx = sx + ga * D[6 + m];
                    Mx[m] = sx;
                    double sy = z * D[m];
                    sy = sy + ga * D[3 + m];
                    sy = sy + ba * D[6 + m];
                    My[m] = sy;
                }
#pragma unroll
                for (int b = 0; b < 3; ++b) { // ascending local column, as the loops of solver.rs:304-322
                    if (nn[b] != j) continue;
                    const double bb = b == 0 ? bd0 : (b == 1 ? bd1 : bd2), gb = b == 0 ? gd0 : (b == 1 ? gd1 : gd2);
                    double t;
                    t = Mx[0] * bb; t = t + Mx[1] * z;  t = t + Mx[2] * gb; k00 += t * area * thick;
                    t = Mx[0] * z;  t = t + Mx[1] * gb; t = t + Mx[2] * bb; k01 += t * area * thick;
                    t = My[0] * bb; t = t + My[1] * z;  t = t + My[2] * gb; k10 += t * area * thick;
                    t = My[0] * z;  t = t + My[1] * gb; t = t + My[2] * bb; k11 += t * area * thick;
                }
            }
        } else { // oversize tile: every block straight from global memory, the k_assemble_rows arithmetic
            for (int32_t q = q0 + s_eoff[lo]; q < q0 + s_eoff[lo + 1]; ++q) {
                const uint32_t v = inc[q];
                const uint32_t e = v / 3u;
                const int a = (int)(v - 3u * e);
                const int32_t n0 = conn[3 * (int64_t)e], n1 = conn[3 * (int64_t)e + 1], n2 = conn[3 * (int64_t)e + 2];
                if (n0 != j && n1 != j && n2 != j) continue;
                const double2 v0 = xy[n0], v1 = xy[n1], v2 = xy[n2];
                const int32_t nn[3] = {n0, n1, n2};
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    if (nn[b] != j) continue;
                    double c00, c01, c10, c11;
                    ke_block(v0, v1, v2, a, b, D, thick, c00, c01, c10, c11);
                    k00 += c00;
                    k01 += c01;
                    k10 += c10;
                    k11 += c11;
                }
            }
        }
        double *r0 = kval + 4 * (int64_t)p + 2 * kpos;
        double *r1 = kval + 4 * (int64_t)p + 2 * cnt + 2 * kpos;
        r0[0] = k00;
        r0[1] = k01;
        r1[0] = k10;
        r1[1] = k11;
    }
}

void assemble_tiles(const int32_t *bcol, const int32_t *bptr, const int32_t *inc_off, const uint32_t *inc,
                    const uint32_t *perm, const int32_t *conn, const double *xy, int64_t N, double nu, double youngs,
                    double thick, double *kval, hipStream_t s)
{
    const int64_t tiles = (N + kAsmNodes - 1) / kAsmNodes;
    k_assemble_tiles<<<(unsigned)tiles, kAsmThreads, 0, s>>>(bcol, bptr, inc_off, inc, perm, conn, (const double2 *)xy, N,
                                                             nu, youngs, thick, kval);
}

// Opt-in preconditioner (the reference has none, solver.rs:142): the node-diagonal 2x2 blocks of
// K, summed over the incident elements in ascending element order with ke_block -- bitwise the diagonal blocks of
// the assembled matrix, so the oracle can rebuild the same M from its own K -- then inverted on the free DOFs and
// rounded to fp32 (oracle/magnetite_oracle.c:orc_block_jacobi, same operations).  One thread per node, Hilbert order.
__global__ void __launch_bounds__(256) k_precond_blocks(const int32_t *inc_off, const uint32_t *inc, const uint32_t *perm,
                                                        const int32_t *conn, const double2 *xy, const uint8_t *u_known,
                                                        int64_t N, double nu, double youngs, double thick, int kind,
                                                        float4 *minvP)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= N) return;
    const int64_t i = perm[g];
    double D[9];
    stress_strain(nu, youngs, D);
    double k00 = 0.0, k01 = 0.0, k11 = 0.0;
    for (int32_t q = inc_off[g]; q < inc_off[g + 1]; ++q) {
        const uint32_t v = inc[q];
        const uint32_t e = v / 3u;
        const int a = (int)(v - 3u * e);
        const int32_t n0 = conn[3 * (int64_t)e], n1 = conn[3 * (int64_t)e + 1], n2 = conn[3 * (int64_t)e + 2];
        double c00, c01, c10, c11;
        ke_block(xy[n0], xy[n1], xy[n2], a, a, D, thick, c00, c01, c10, c11);
        k00 += c00;
        k01 += c01;
        k11 += c11;
    }
    const bool fx = !u_known[2 * i], fy = !u_known[2 * i + 1];
    double i00 = 0.0, i01 = 0.0, i11 = 0.0;
    if (fx && fy && kind == 2) {
        const double det = k00 * k11 - k01 * k01;
        i00 = k11 / det;
        i01 = (k01 / det) * -1.0;
        i11 = k00 / det;
    } else {
        if (fx) i00 = 1.0 / k00;
        if (fy) i11 = 1.0 / k11;
    }
    minvP[g] = make_float4((float)i00, (float)i01, (float)i11, 0.f);
}

void precond_blocks(const int32_t *inc_off, const uint32_t *inc, const uint32_t *perm, const int32_t *conn,
                    const double *xy, const uint8_t *u_known, int64_t N, double nu, double youngs, double thick, int kind,
                    float4 *minvP, hipStream_t s)
{
    k_precond_blocks<<<blocks_for(N, 256), 256, 0, s>>>(inc_off, inc, perm, conn, (const double2 *)xy, u_known, N, nu,
                                                        youngs, thick, kind, minvP);
}

// solver.rs:365-404 + 427-432 on the CSR rows: known[r,k] = -(K[r,col]*u[col]) summed ascending, + f.
__device__ inline double rhs_row(const int32_t *bptr, const int32_t *bcol, const double *kval, const uint8_t *u_known,
                                 const double *u_in, const double *f_in, int64_t r)
{
    const int64_t i = r >> 1;
    const int a = (int)(r & 1);
    const int32_t p = bptr[i], nb = bptr[i + 1] - p;
    const double *row = kval + 4 * (int64_t)p + (int64_t)a * 2 * nb;
    double s = 0.0;
    for (int k = 0; k < nb; ++k) {
        const int64_t j = bcol[p + k];
        if (u_known[2 * j]) s += (row[2 * k] * u_in[2 * j]) * -1.0;
        if (u_known[2 * j + 1]) s += (row[2 * k + 1] * u_in[2 * j + 1]) * -1.0;
    }
    return s + f_in[r];
}

__global__ void __launch_bounds__(256) k_rhs_from_csr(const int32_t *bptr, const int32_t *bcol, const double *kval,
                                                      const uint8_t *u_known, const double *u_in, const double *f_in,
                                                      const int32_t *iperm, int64_t N, double *bP)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= 2 * N) return;
    const double v = u_known[r] ? 0.0 : rhs_row(bptr, bcol, kval, u_known, u_in, f_in, r);
    bP[2 * (int64_t)iperm[r >> 1] + (r & 1)] = v;
}

void rhs_from_csr(const int32_t *bptr, const int32_t *bcol, const double *kval, const uint8_t *u_known,
                  const double *u_in, const double *f_in, const int32_t *iperm, int64_t N, double *bP, hipStream_t s)
{
    k_rhs_from_csr<<<blocks_for(2 * N, 256), 256, 0, s>>>(bptr, bcol, kval, u_known, u_in, f_in, iperm, N, bP);
}

__global__ void __launch_bounds__(256) k_rhs_compact(const int32_t *bptr, const int32_t *bcol, const double *kval,
                                                     const uint8_t *u_known, const double *u_in, const double *f_in,
                                                     const int32_t *fidx, int64_t N, double *b)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= 2 * N || u_known[r]) return;
    b[fidx[r]] = rhs_row(bptr, bcol, kval, u_known, u_in, f_in, r);
}

void rhs_compact(const int32_t *bptr, const int32_t *bcol, const double *kval, const uint8_t *u_known,
                 const double *u_in, const double *f_in, const int32_t *fidx, int64_t N, double *b, hipStream_t s)
{
    k_rhs_compact<<<blocks_for(2 * N, 256), 256, 0, s>>>(bptr, bcol, kval, u_known, u_in, f_in, fidx, N, b);
}

// solver.rs:443-454
__global__ void __launch_bounds__(256) k_scatter_back(const double2 *xP, const uint32_t *perm, const uint8_t *u_known,
                                                      const double *u_in, int64_t N, double *u)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int64_t o = perm[i];
    const double2 x = xP[i];
    u[2 * o] = u_known[2 * o] ? u_in[2 * o] : x.x;
    u[2 * o + 1] = u_known[2 * o + 1] ? u_in[2 * o + 1] : x.y;
}

void scatter_back(const double *xP, const uint32_t *perm, const uint8_t *u_known, const double *u_in, int64_t N,
                  double *u, hipStream_t s)
{
    k_scatter_back<<<blocks_for(N, 256), 256, 0, s>>>((const double2 *)xP, perm, u_known, u_in, N, u);
}

// solver.rs:456-469: full row . u in ascending column order (structural zeros of the dense row add +-0)
__global__ void __launch_bounds__(256) k_reactions_from_csr(const int32_t *bptr, const int32_t *bcol,
                                                            const double *kval, const uint8_t *u_known,
                                                            const double *u, const double *f_in, int64_t N, double *f)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= 2 * N) return;
    if (!u_known[r]) {
        f[r] = f_in[r];
        return;
    }
    const int64_t i = r >> 1;
    const int a = (int)(r & 1);
    const int32_t p = bptr[i], nb = bptr[i + 1] - p;
    const double *row = kval + 4 * (int64_t)p + (int64_t)a * 2 * nb;
    double s = 0.0;
    for (int k = 0; k < nb; ++k) {
        const int64_t j = bcol[p + k];
        s += row[2 * k] * u[2 * j];
        s += row[2 * k + 1] * u[2 * j + 1];
    }
    f[r] = s;
}

void reactions_from_csr(const int32_t *bptr, const int32_t *bcol, const double *kval, const uint8_t *u_known,
                        const double *u, const double *f_in, int64_t N, double *f, hipStream_t s)
{
    k_reactions_from_csr<<<blocks_for(2 * N, 256), 256, 0, s>>>(bptr, bcol, kval, u_known, u, f_in, N, f);
}

__global__ void __launch_bounds__(256) k_reactions_from_apply(const double *yP, const int32_t *iperm,
                                                              const uint8_t *u_known, const double *f_in, int64_t N,
                                                              double *f)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= 2 * N) return;
    f[r] = u_known[r] ? yP[2 * (int64_t)iperm[r >> 1] + (r & 1)] : f_in[r];
}

void reactions_from_apply(const double *yP, const int32_t *iperm, const uint8_t *u_known, const double *f_in,
                          int64_t N, double *f, hipStream_t s)
{
    k_reactions_from_apply<<<blocks_for(2 * N, 256), 256, 0, s>>>(yP, iperm, u_known, f_in, N, f);
}

// solver.rs:496-535: sigma = (D*B)*u_e; scalar = sqrt(sx^2+sy^2) * (sx+sy < 1.0 ? -1 : 1) -- quirk kept.
__global__ void __launch_bounds__(256) k_element_stress(const double2 *xy, const int32_t *conn, const double2 *u,
                                                        int64_t E, double nu, double youngs, double *stress)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int32_t n0 = conn[3 * e], n1 = conn[3 * e + 1], n2 = conn[3 * e + 2];
    const double2 v0 = xy[n0], v1 = xy[n1], v2 = xy[n2];
    const double2 u0 = u[n0], u1 = u[n1], u2 = u[n2];
    const double ue[6] = {u0.x, u0.y, u1.x, u1.y, u2.x, u2.y};
    double D[9], B[18], DB[18], sg[3];
    stress_strain(nu, youngs, D);
    strain_displacement(v0.x, v0.y, v1.x, v1.y, v2.x, v2.y, signed_area(v0.x, v0.y, v1.x, v1.y, v2.x, v2.y), B);
    matmul<3, 3, 6>(D, B, DB);
    matmul<3, 6, 1>(DB, ue, sg);
    const double sign = (sg[0] + sg[1] < 1.0) ? -1.0 : 1.0;
    stress[e] = sqrt(sg[0] * sg[0] + sg[1] * sg[1]) * sign;
}

void element_stress(const double *xy, const int32_t *conn, const double *u, int64_t E, double nu, double youngs,
                    double *stress, hipStream_t s)
{
    k_element_stress<<<blocks_for(E, 256), 256, 0, s>>>((const double2 *)xy, conn, (const double2 *)u, E, nu, youngs,
                                                        stress);
}

} // namespace magk
