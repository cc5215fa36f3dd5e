// Reference-order fp64 arithmetic of the path, one kernel per reference
// function.  Compiled with -ffp-contract=off: the Rust reference never fuses
// a*b+c, and nalgebra's static matrix product sums ascending k starting from
// the k=0 product (gemm -> gemv -> axcpy), so K_e, the assembled K, the RHS
// and the stress scalar reproduce the reference's rounding, not just its
// formulas.  None of these kernels is in the CG loop.
#include <algorithm>
#include <cstdlib>
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
// bit-exact.  A tile is kAsmNodes consecutive nodes of the Hilbert order -- a compact patch of the mesh -- with every
// element incident on them; one workgroup (two wavefronts) per tile, one thread per ROW NODE i:
//   * the thread walks node i's incidence entries (element e, corner a) in ascending element order -- the reference's
//     `+=` order (solver.rs:299-325) -- kAsmBatch at a time: their connectivity triples are gathered together, then
//     their coordinates, so a thread keeps 16-24 loads in flight instead of one dependent chain per element (measured:
//     the gathers, not the arithmetic, set this kernel's time);
//   * per entry the element is evaluated ONCE: signed area, the seven quotients of `strain_displacement_mat /= 2.0 *
//     area` (solver.rs:187-230: every true fp64 division of K_e), rows 2a and 2a+1 of B^T D -- and from them the three
//     2x2 blocks ((B^T D) B)[2a..][2b..] * area * thickness, b = 0, 1, 2, each entry with exactly nalgebra's operations
//     (ascending-k sums from the first product);
//   * the blocks are added into the row's accumulators, which live in LDS ([slot][thread], 16-byte pieces: conflict-
//     free), at the position of node n_b among the row's sorted columns; at the end the thread stores its two scalar
//     CSR rows (two runs of 16 * cnt bytes).
// Against k_assemble_rows (one thread per block re-deriving area and five divisions per matching element, each row's
// entries scanned by every block of the row: ~9 evaluations and ~21 entry visits per element) an element is evaluated
// once per incident node (3) and visited 3 times.  Rows with more than kAsmSlots blocks (valence > 7) do not fit the
// accumulators: the workgroup finishes them together, one thread per block, with the k_assemble_rows arithmetic --
// same bits.
constexpr int kAsmNodes = 128; // row nodes (threads) per tile
constexpr int kAsmSlots = 8;   // 2x2 blocks a row may hold in its LDS accumulators (valence 6 interior node: 7)
constexpr int kAsmBatch = 4;   // incidence entries whose gathers a thread issues together (the kernel is latency-bound)

template <int STRIDE = kAsmNodes>
__device__ inline void asm_entry(const int32_t (&nn)[3], int a, const double2 v0, const double2 v1, const double2 v2,
                                 const double *D, double thick, const int32_t (&col)[kAsmSlots], double2 *s_top,
                                 double2 *s_bot, int lane)
{
    const double area = signed_area(v0.x, v0.y, v1.x, v1.y, v2.x, v2.y);
    const double d = 2.0 * area;
    const double bd[3] = {(v1.y - v2.y) / d, (v2.y - v0.y) / d, (v0.y - v1.y) / d};
    const double gd[3] = {(v2.x - v1.x) / d, (v0.x - v2.x) / d, (v1.x - v0.x) / d};
    const double z = 0.0 / d; // the structural zeros of B after the division
    const double ba = a == 0 ? bd[0] : (a == 1 ? bd[1] : bd[2]), ga = a == 0 ? gd[0] : (a == 1 ? gd[1] : gd[2]);
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
        const double bb = bd[b], gb = gd[b];
        double t, c00, c01, c10, c11;
        t = Mx[0] * bb; t = t + Mx[1] * z;  t = t + Mx[2] * gb; c00 = t * area * thick;
        t = Mx[0] * z;  t = t + Mx[1] * gb; t = t + Mx[2] * bb; c01 = t * area * thick;
        t = My[0] * bb; t = t + My[1] * z;  t = t + My[2] * gb; c10 = t * area * thick;
        t = My[0] * z;  t = t + My[1] * gb; t = t + My[2] * bb; c11 = t * area * thick;
        int kpos = 0; // position of n_b among the row's ascending columns (it is one of them)
#pragma unroll
        for (int k = 0; k < kAsmSlots; ++k) kpos += col[k] < nn[b] ? 1 : 0;
        double2 top = s_top[kpos * STRIDE + lane], bot = s_bot[kpos * STRIDE + lane];
        top.x += c00;
        top.y += c01;
        bot.x += c10;
        bot.y += c11;
        s_top[kpos * STRIDE + lane] = top;
        s_bot[kpos * STRIDE + lane] = bot;
    }
}

__global__ void __launch_bounds__(kAsmNodes) k_assemble_tiles(const int32_t *bcol, const int32_t *bptr,
                                                               const int32_t *inc_off, const uint32_t *inc,
                                                               const uint32_t *perm, const int32_t *conn,
                                                               const double2 *xy, int64_t N, double nu, double youngs,
                                                               double thick, double *kval)
{
    __shared__ double2 s_top[kAsmSlots * kAsmNodes]; // (k00, k01) of block `slot` of the thread's row
    __shared__ double2 s_bot[kAsmSlots * kAsmNodes]; // (k10, k11)
    const int lane = threadIdx.x;
    const int64_t g = (int64_t)blockIdx.x * kAsmNodes + lane;
    double D[9];
    stress_strain(nu, youngs, D);
    int64_t i = -1;
    int32_t p = 0, cnt = 0;
    if (g < N) {
        i = perm[g];
        p = bptr[i];
        cnt = bptr[i + 1] - p;
    }
    // cnt == 0: a row this rank does not keep (several GPUs: csr_symbolic filtered its pairs out); every kept row
    // holds at least its diagonal block
    const bool fast = g < N && cnt > 0 && cnt <= kAsmSlots;
    if (fast) {
        int32_t col[kAsmSlots];
#pragma unroll
        for (int k = 0; k < kAsmSlots; ++k) {
            col[k] = k < cnt ? bcol[p + k] : 0x7fffffff;
            s_top[k * kAsmNodes + lane] = make_double2(0.0, 0.0);
            s_bot[k * kAsmNodes + lane] = make_double2(0.0, 0.0);
        }
        const int32_t q1 = inc_off[g + 1];
        for (int32_t q = inc_off[g]; q < q1; q += kAsmBatch) {
            uint32_t v[kAsmBatch];
            int32_t nn[kAsmBatch][3];
            double2 c[kAsmBatch][3];
#pragma unroll
            for (int u = 0; u < kAsmBatch; ++u) v[u] = q + u < q1 ? inc[q + u] : 0xffffffffu;
#pragma unroll
            for (int u = 0; u < kAsmBatch; ++u)
                if (v[u] != 0xffffffffu) {
                    const int3 t = ((const int3 *)conn)[v[u] / 3u]; // one 12-byte load: a third of the address traffic
                    nn[u][0] = t.x;
                    nn[u][1] = t.y;
                    nn[u][2] = t.z;
                }
#pragma unroll
            for (int u = 0; u < kAsmBatch; ++u)
                if (v[u] != 0xffffffffu) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) c[u][k] = xy[nn[u][k]];
                }
#pragma unroll
            for (int u = 0; u < kAsmBatch; ++u)
                if (v[u] != 0xffffffffu)
                    asm_entry(nn[u], (int)(v[u] % 3u), c[u][0], c[u][1], c[u][2], D, thick, col, s_top, s_bot, lane);
        }
        double *r0 = kval + 4 * (int64_t)p, *r1 = r0 + 2 * cnt;
#pragma unroll
        for (int k = 0; k < kAsmSlots; ++k)
            if (k < cnt) {
                const double2 top = s_top[k * kAsmNodes + lane], bot = s_bot[k * kAsmNodes + lane];
                r0[2 * k] = top.x;
                r0[2 * k + 1] = top.y;
                r1[2 * k] = bot.x;
                r1[2 * k + 1] = bot.y;
            }
    }
    // rows that did not fit the accumulators: the whole workgroup, one thread per block, k_assemble_rows arithmetic
    if (!__syncthreads_or(g < N && cnt > kAsmSlots)) return;
    int32_t *s_big = (int32_t *)s_top; // every fast row has been stored: the accumulators are free (5 workgroups of
    s_big[lane] = (g < N && cnt > kAsmSlots) ? (int32_t)i : -1; // exactly 32 KiB fit a CU's 160 KiB)
    __syncthreads();
    for (int m = 0; m < kAsmNodes; ++m) {
        const int32_t ib = s_big[m];
        if (ib < 0) continue; // uniform
        const int64_t gb_ = (int64_t)blockIdx.x * kAsmNodes + m;
        const int32_t pb = bptr[ib], cb = bptr[ib + 1] - pb;
        for (int kpos = lane; kpos < cb; kpos += kAsmNodes) {
            const int32_t j = bcol[pb + kpos];
            double k00 = 0.0, k01 = 0.0, k10 = 0.0, k11 = 0.0;
            for (int32_t q = inc_off[gb_]; q < inc_off[gb_ + 1]; ++q) {
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
            double *r0 = kval + 4 * (int64_t)pb + 2 * kpos;
            double *r1 = kval + 4 * (int64_t)pb + 2 * cb + 2 * kpos;
            r0[0] = k00;
            r0[1] = k01;
            r1[0] = k10;
            r1[1] = k11;
        }
    }
}

void assemble_tiles(const int32_t *bcol, const int32_t *bptr, const int32_t *inc_off, const uint32_t *inc,
                    const uint32_t *perm, const int32_t *conn, const double *xy, int64_t N, double nu, double youngs,
                    double thick, double *kval, hipStream_t s)
{
    const int64_t tiles = (N + kAsmNodes - 1) / kAsmNodes;
    k_assemble_tiles<<<(unsigned)tiles, kAsmNodes, 0, s>>>(bcol, bptr, inc_off, inc, perm, conn, (const double2 *)xy, N,
                                                           nu, youngs, thick, kval);
}

// ---- the same assembly fed from the CG TILES (round 3; the shape BASELINE.json's north star names: "LDS staging of node
// coordinates and coalesced HBM reads of the element->node index arrays") -------------------------------------------
// k_assemble_tiles walks  incidence entry -> connectivity triple -> three coordinates  in caller numbering (three
// dependent gathers per (node, element), 4.7 x the algorithmic read traffic), every thread loads its row's columns and
// stores its row's blocks by itself (each of those wave instructions touches 64 different cache lines; PMC: L1 stalled on
// pending requests 57 % of the launch, 128 MB written for 112 MB of K), and its accumulators -- 256 bytes of LDS per row
// node -- cap the CU at ten waves, too few to cover any of that.  Measured in round 3: feeding the same
// thread-per-row kernel from the CG tiles and storing row-wise brought 76 -> 70 us; what remained was the latency of a
// workgroup's serial phases with two waves per SIMD.  This kernel drops the accumulators:
//
//   ONE LANE PER (row node a, incident triangle (a, b, c)), eight lanes per node -- a wave holds eight nodes (six lanes and
//   ten nodes in tiles where no node has more than six triangles: the rule on structured meshes).
//
// The ordering phase has already built, per tile of B consecutive Hilbert nodes, what the gathers look up: the tile's owned
// + halo coordinates as two contiguous runs (xyP, halo_xy) and, per node, one word per incident element in ascending
// element order -- the two OTHER corners as tile-local ids (symbolic.hip, k_fill_ell16; kept for this kernel with the
// node's corner label in the spare bits: lb | label << 12 | lc << 16, orientation a -> b -> c of the reference's vertex
// order) plus, in a 16-bit companion word, WHERE the triangle's blocks go: the positions of b, of c and of the diagonal
// among the row's columns (ascending caller id), counted once per mesh by the symbolic phase -- the kernel reads neither
// the column indices nor any caller id but its own nodes'.  A lane evaluates its triangle once (asm_fan_blocks: the rotated
// frame; only the ORDER in which solver.rs:187-193 sums the three products of the signed area depends on the label) and
// holds the three blocks of row a it contributes to: the diagonal's share, the block of column b, the block of column c.
// Then, per row:
//   * an off-diagonal block (a, n) receives at most two contributions in a manifold mesh -- from the triangle that has n
//     as its b and the one that has it as its c, the two sides of the edge a-n.  The reference adds them in ascending
//     element order, (0.0 + c1) + c2; IEEE addition is commutative (and (0.0 + x) + y = (0.0 + y) + x also when signed
//     zeros are involved), so the b-side lane fetches its partner's c-side block from LDS and forms (0.0 + own) + partner's;
//     a c with no b-side partner (the open end of a boundary fan) is finished by its own lane;
//   * the diagonal block receives one contribution per triangle and there the order does matter: the lanes of a node hold
//     their triangles in ascending element order (slot order of the table); four lanes of the node add up one entry of the
//     block each, from LDS, in that order, starting from 0.0;
//   * the finished 16-byte pieces are put where they belong in a staging copy of the node's run of K (its two rows are one
//     contiguous run of 32 * cnt bytes) and the wave writes the runs out sixteen lanes per node: whole runs per store
//     instruction.  (Stored piece by piece -- one 16-byte piece per triangle and instruction -- the kernel was bound by its
//     3 M write requests of 37 bytes: same 48 us with the arithmetic removed, 36 us with the stores removed.)
// No accumulators: the image (16 bytes per staged node, 4 per owned node) and 64 bytes per lane of exchange / staging
// space make ~31 KB of LDS per 256-thread workgroup -- four workgroups (16 waves) per CU with the kernel's 110 VGPRs.  The seven quotients of `B /= 2A` share their denominator:
// with r = RN(1 / 2A) (one true division) each quotient is x r corrected twice with exact FMA residuals (Markstein's
// sequence, the one IA-64 and POWER divide with): RN(x / 2A), the reference's bits, in 5 instructions instead of a
// ~13-instruction division each.  Valid while nothing under- or overflows and 2A's significand is not all ones:
// guaranteed per tile by a range check of the staged coordinates (0 or 1e-150 < |c| < 1e30, so a non-zero difference is
// >= 2e-166) and per element by 1e-40 < |2A| < 1e60; otherwise the true divisions run.  Under the same bounds (and
// |D| < 1e60, |thickness| < 1e30) the terms with a structural zero of B or D are left out (see asm_fan_blocks).
// Rows the scheme does not cover -- more than 8 triangles, an edge with three or more triangles, an element
// that lists a node twice -- are finished by the whole workgroup with the k_assemble_rows arithmetic (label order, true
// divisions), as in k_assemble_tiles.  Tiles whose image does not fit fall back to k_assemble_tiles altogether.
__device__ inline double div_shared(double x, double d, double r)
{
    double q = x * r;
    double e = __builtin_fma(-d, q, x);
    q = __builtin_fma(e, r, q);
    e = __builtin_fma(-d, q, x);
    return __builtin_fma(e, r, q);
}

__device__ inline bool coord_in_range(double c)
{
    const double a = fabs(c);
    return c == 0.0 || (a > 1e-150 && a < 1e30);
}

// The three 2x2 blocks of row a from triangle (a, b, c), the three nodes distinct, in the rotated frame: dg = K_e[a][a],
// kb = K_e[a][b], kc = K_e[a][c], each entry with the reference's operations ((B^T D) B, ascending-k sums from the first
// product, * area * thickness).  What depends on the label of a is only the order of the area's three products:
// label order is (a, b, c) for label 0, (c, a, b) for label 1, (b, c, a) for label 2.
__device__ inline void asm_fan_blocks(int a, const double2 va, const double2 vb, const double2 vc, const double *D,
                                      double thick, bool sane, double (&dg)[4], double (&kb)[4], double (&kc)[4])
{
    const double nba = vb.y - vc.y, nbb = vc.y - va.y, nbc = va.y - vb.y; // numerators of beta_a, beta_b, beta_c
    const double nga = vc.x - vb.x, ngb = va.x - vc.x, ngc = vb.x - va.x; // ... of gamma_a, gamma_b, gamma_c
    const double Ta = va.x * nba, Tb = vb.x * nbb, Tc = vc.x * nbc;
    const double t1 = a == 0 ? Ta : (a == 1 ? Tc : Tb);
    const double t2 = a == 0 ? Tb : (a == 1 ? Ta : Tc);
    const double t3 = a == 0 ? Tc : (a == 1 ? Tb : Ta);
    const double area = 0.5 * (t1 + t2 + t3);
    const double d = 2.0 * area;
    double ba, bb, bc, ga, gb, gc, z;
    const unsigned long long dbits = (unsigned long long)__double_as_longlong(d);
    const bool quick = sane && fabs(d) > 1e-40 && fabs(d) < 1e60 &&
                       (dbits & 0x000fffffffffffffull) != 0x000fffffffffffffull;
    if (quick) {
        const double r = 1.0 / d; // correctly rounded: the one true division
        ba = div_shared(nba, d, r);
        bb = div_shared(nbb, d, r);
        bc = div_shared(nbc, d, r);
        ga = div_shared(nga, d, r);
        gb = div_shared(ngb, d, r);
        gc = div_shared(ngc, d, r);
        // `sane` bounds every factor (|coordinates| < 1e30, |D| < 1e60, |thickness| < 1e30) and here |2A| > 1e-40, so
        // nothing below overflows: quotients < 2e70, B^T D < 2e130, the sums < 8e200, * area * thickness < 4e290.  Then
        // every term the reference forms with a structural zero of B (0.0 / 2A) or of D (0.0 * E / (1 - nu^2)) is +-0, and
        // x + (+-0) = x for x != 0: leaving those terms out can only change the SIGN of an intermediate zero, which no
        // later operation turns into anything but a zero, and the scatter's `0.0 + c` makes that +0.0 either way.  What
        // stays of rows 2a, 2a+1 of B^T D is (ba D00, ba D01, ga D22) and (ga D10, ga D11, ba D22), and of an entry the two
        // products without a zero factor, in the reference's order.
        const double Mx0 = ba * D[0], Mx1 = ba * D[1], Mx2 = ga * D[8];
        const double My0 = ga * D[3], My1 = ga * D[4], My2 = ba * D[8];
        auto block = [&](double bn, double gn, double (&c)[4]) {
            c[0] = (Mx0 * bn + Mx2 * gn) * area * thick;
            c[1] = (Mx1 * gn + Mx2 * bn) * area * thick;
            c[2] = (My0 * bn + My2 * gn) * area * thick;
            c[3] = (My1 * gn + My2 * bn) * area * thick;
        };
        block(ba, ga, dg);
        block(bb, gb, kb);
        block(bc, gc, kc);
        return;
    }
    ba = nba / d;
    bb = nbb / d;
    bc = nbc / d;
    ga = nga / d;
    gb = ngb / d;
    gc = ngc / d;
    z = 0.0 / d; // the structural zeros of B after the division
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
    auto block = [&](double bn, double gn, double (&c)[4]) {
        double t;
        t = Mx[0] * bn; t = t + Mx[1] * z;  t = t + Mx[2] * gn; c[0] = t * area * thick;
        t = Mx[0] * z;  t = t + Mx[1] * gn; t = t + Mx[2] * bn; c[1] = t * area * thick;
        t = My[0] * bn; t = t + My[1] * z;  t = t + My[2] * gn; c[2] = t * area * thick;
        t = My[0] * z;  t = t + My[1] * gn; t = t + My[2] * bn; c[3] = t * area * thick;
    };
    block(ba, ga, dg);
    block(bb, gb, kb);
    block(bc, gc, kc);
}

__device__ inline void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// One 16-byte piece of K.  MAG_ASM_STORE: 0 plain (write-back), 1 non-temporal, 2 write-through (sc1: the bytes go out while
// the kernel still computes).  Measured in one session, numeric assembly at 1M / 4M triangles: 45.0 / 166.8 us plain,
// 42.9 / 171.2 non-temporal, 42.8 / 162.4 write-through (scripts/asm_store_ab.py): write-through.
#ifndef MAG_ASM_STORE
#define MAG_ASM_STORE 2
#endif
__device__ inline void k_store(double *dst, double2 v)
{
#if MAG_ASM_STORE == 1
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d t = {v.x, v.y};
    __builtin_nontemporal_store(t, (v2d *)dst);
#elif MAG_ASM_STORE == 2
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    u4 t;
    __builtin_memcpy(&t, &v, 16);
    // (s_nop: the registers of a store of more than 8 bytes must not be rewritten in the next two cycles; the compiler
    // keeps that distance for its own stores and cannot see into this one)
    asm volatile("global_store_dwordx4 %0, %1, off sc1" MAG_WS_DATA : : "v"(dst), "v"(t) : "memory");
#else
    *(double2 *)dst = v;
#endif
}

constexpr int kFanThreads = 256; // 32 nodes x 8 lanes per pass
constexpr int kFanLanes = 8;     // lanes (triangles, columns) per node
constexpr int kFanStage = 20;    // 16-byte pieces of staging per node: two rows of at most ten blocks

template <int B>
__global__ void __launch_bounds__(kFanThreads) k_assemble_fan(const int32_t *bcol, const int32_t *bptr, const uint32_t *perm,
                                                              const double2 *xyP, const double2 *halo_xy,
                                                              const int32_t *tile_hoff, const int32_t *tile_deg,
                                                              const int64_t *tile_off, const uint32_t *ell_asm,
                                                              const uint16_t *ell_pos, const int32_t *inc_off,
                                                              const uint32_t *inc, const int32_t *conn, const double2 *xy,
                                                              int64_t N, int32_t cap, int32_t img_pieces, int32_t segs,
                                                              int32_t six, double nu, double youngs, double thick, double *kval)
{
    extern __shared__ __attribute__((aligned(16))) double2 s_asm[];
    double2 *s_xy = s_asm;                                   // [cap] coordinates of the tile's owned + halo nodes
    int32_t *s_cid = (int32_t *)(s_xy + cap);                // [B] caller ids of the owned nodes
    double2 *s_cc = s_xy + img_pieces;                       // [2][256] the lanes' c-side blocks (top piece, bottom piece)
    double2 *s_dd = s_cc + 2 * kFanThreads;                  // [2][256] the lanes' diagonal shares
    int32_t *s_big = (int32_t *)(s_dd + 2 * kFanThreads);    // [B] rows left to the workgroup: caller id, or -1
    const int lane = threadIdx.x;
    // `segs` workgroups share a tile (each stages the whole image and takes B / segs of its row nodes): meshes of few
    // tiles still fill the chip
    const int32_t t = blockIdx.x / segs, seg = blockIdx.x % segs;
    const int32_t lfirst = seg * (B / segs), lend = lfirst + B / segs;
    // ---- the tile's image: coordinates of its owned and halo nodes, caller ids of the owned ones
    const int32_t hoff = tile_hoff[t], nh = tile_hoff[t + 1] - hoff;
    bool in_range = true;
    for (int32_t l = lane; l < B; l += kFanThreads) {
        const int64_t gg = (int64_t)t * B + l;
        s_big[l] = -1;
        if (gg < N) {
            const double2 c = xyP[gg];
            s_xy[l] = c;
            s_cid[l] = (int32_t)perm[gg];
            in_range &= coord_in_range(c.x) && coord_in_range(c.y);
        }
    }
    for (int32_t h = lane; h < nh; h += kFanThreads) {
        const double2 c = halo_xy[hoff + h];
        s_xy[B + h] = c;
        in_range &= coord_in_range(c.x) && coord_in_range(c.y);
    }
    double D[9];
    stress_strain(nu, youngs, D);
#pragma unroll
    for (int m = 0; m < 9; ++m) in_range &= fabs(D[m]) < 1e60; // false for NaN as well
    in_range &= fabs(thick) < 1e30;
    const bool sane = __syncthreads_and(in_range ? 1 : 0) != 0;
    const int32_t td = tile_deg[t];
    const uint32_t *table = ell_asm + tile_off[t];
    const uint16_t *ptable = ell_pos + tile_off[t];
    bool any_big = false;
    // Software pipeline over the passes: the row pointers and the two table words of pass n + 1 are requested at the top of
    // pass n -- unconditionally, from a clamped node, so that no branch sits between a load and its first use (the
    // compiler drains the memory counter at every join a load crosses: measured, the conditional form of this fetch waited
    // for its own loads on the spot and for the previous pass's stores with them, ~3 us per pass for 0.4 us of
    // arithmetic) -- and they are collected after the evaluation, BEFORE this pass's stores are issued: the counter is
    // in-order, so a wait placed after the stores would wait for their acknowledgements too.
    struct Row {
        int32_t i, p, pe;
        uint32_t w, pos;
    };
    const int32_t nvalid = (int32_t)((N - (int64_t)t * B) < (int64_t)B ? (N - (int64_t)t * B) : (int64_t)B);
    const int32_t llast = (lend < nvalid ? lend : nvalid) - 1; // last node of this segment that exists
    // LANES lanes per node: eight in general; six when no node of the tile has more triangles (the rule on structured
    // meshes: a closed fan of valence 6) -- ten nodes per wave instead of eight, four idle lanes instead of a quarter.
    auto run = [&](auto lanes_c) {
        constexpr int LANES = decltype(lanes_c)::value, NPW = 64 / LANES; // nodes per wave
        constexpr int kStep = (kFanThreads / 64) * NPW;                   // nodes per pass
        const int nw = (lane & 63) / LANES, k = (lane & 63) - nw * LANES, gbase = lane - k;
        const bool lane_used = nw < NPW;
        const int wl = lane & 63, gb = gbase & 63;
        double2 *wreg = s_cc + (lane >> 6) * 256; // this wave's exchange area: 4 x 64 pieces (s_cc and s_dd are contiguous)
        int2 *hdr = (int2 *)(wreg + NPW * kFanStage); // per node of the wave: start of its run of K, blocks in its rows
        const int32_t lnode = (lane >> 6) * NPW + nw; // this lane's node inside a pass
        const int kk = k < td ? k : td - 1; // every word of a node carries the diagonal's position: lanes beyond the tile's
                                            // row length (they sum diagonal entries, see below) read it from the last slot
        auto fetch = [&](int32_t l) {
            const int32_t lcl = l <= llast ? l : llast;
            Row r;
            r.i = s_cid[lcl];
            r.p = bptr[r.i];
            r.pe = bptr[r.i + 1];
            r.w = table[(int64_t)kk * B + lcl];
            r.pos = ptable[(int64_t)kk * B + lcl];
            return r;
        };
        Row cur = fetch(lfirst + lnode);
        __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): the loop is entered with nothing in flight, as it is re-entered
        for (int32_t l0 = lfirst; l0 < lend; l0 += kStep) {
            const int32_t l = l0 + lnode; // this lane's node inside the tile
            const Row nxt = fetch(l + kStep);   // pass n + 1: in flight during the evaluation below
            const bool valid = l <= llast && lane_used;
            const int32_t i = valid ? cur.i : -1, p = cur.p;
            const int32_t cnt = valid ? cur.pe - cur.p : 0; // 0: a row this rank does not keep (several GPUs)
            const uint32_t w = (valid && k < td) ? cur.w : 0xffffffffu, pos = cur.pos;
            const bool live = w != 0xffffffffu;
            // bit 15: the node cannot be assembled triangle by triangle (every word of the node carries it)
            const bool fanrow = live && cnt > 0 && !((w >> 15) & 1u);
            const uint32_t lb = live ? (w & 0xfffu) : 0u, lc = live ? ((w >> 16) & 0xfffu) : 0u;
            double dgc[4] = {0.0, 0.0, 0.0, 0.0}, kbv[4] = {0.0, 0.0, 0.0, 0.0}, kcv[4] = {0.0, 0.0, 0.0, 0.0};
            if (fanrow) {
                asm_fan_blocks((int)((w >> 12) & 3u), s_xy[l], s_xy[lb], s_xy[lc], D, thick, sane, dgc, kbv, kcv);
                wreg[wl] = make_double2(kcv[0], kcv[1]); // the c-side block: top piece, bottom piece
                wreg[64 + wl] = make_double2(kcv[2], kcv[3]);
            }
            wreg[128 + wl] = make_double2(dgc[0], dgc[1]); // every lane: the diagonal's sum runs over all the node's lanes
            wreg[192 + wl] = make_double2(dgc[2], dgc[3]);
            // the node's first lane holds its first triangle: a row with any block has one, and it tells whether the row is a fan
            const bool rowfan = ((__ballot(fanrow ? 1 : 0) >> gb) & 1ull) != 0;
            wave_lds_sync();
            // positions of b and c among the row's ascending columns: from the symbolic phase (k_fill_ell16)
            const int kb_pos = (int)(pos & 15u), kc_pos = (int)((pos >> 4) & 15u);
            double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
            if (fanrow) {
                t0 = 0.0 + kbv[0], t1 = 0.0 + kbv[1], t2 = 0.0 + kbv[2], t3 = 0.0 + kbv[3];
                if (w >> 31) { // the other side of the edge a-b: the triangle whose c is this one's b
                    const int pj = gb + (int)((w >> 28) & 7u);
                    const double2 q0 = wreg[pj], q1 = wreg[64 + pj];
                    t0 = t0 + q0.x;
                    t1 = t1 + q0.y;
                    t2 = t2 + q1.x;
                    t3 = t3 + q1.y;
                }
            }
            double dsum = 0.0;
            if (rowfan && k < 4) {
                // the diagonal: the triangles' shares in ascending element order (slot order), from 0.0; lane k of the
                // node sums entry k of the block.  Lanes without a triangle left +0.0 in their slots, and a sum that
                // starts from +0.0 never becomes -0.0 (x + y = -0.0 only for x = y = -0.0): their shares change nothing.
                const double *dd = (const double *)(wreg + 128 + (k >> 1) * 64 + gb) + (k & 1);
#pragma unroll
                for (int j = 0; j < LANES; ++j) dsum += dd[2 * j];
            }
            // ---- the finished pieces go to the wave's staging area, laid out like the node's run of K: 2 cnt pieces of
            // 16 bytes (row 2i: cnt blocks' top halves, row 2i + 1: their bottom halves).  It takes the exchange area's
            // place: a wave's LDS operations execute in order, its reads above are served before these writes.
            wave_lds_sync();
            double2 *stage = wreg + nw * kFanStage;
            if (fanrow) {
                stage[kb_pos] = make_double2(t0, t1);
                stage[cnt + kb_pos] = make_double2(t2, t3);
                if ((w >> 14) & 1u) { // c has no b-side triangle: the open end of a boundary fan
                    stage[kc_pos] = make_double2(0.0 + kcv[0], 0.0 + kcv[1]);
                    stage[cnt + kc_pos] = make_double2(0.0 + kcv[2], 0.0 + kcv[3]);
                }
            }
            if (rowfan && k < 4) ((double *)(stage + (k >> 1) * cnt + (int)((pos >> 8) & 15u)))[k & 1] = dsum;
            if (k == 0 && lane_used) hdr[nw] = make_int2(p, rowfan ? cnt : 0);
            wave_lds_sync();
            // ---- the next pass's words are here by now (requested before the evaluation); collect them, then store
            __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
            cur = nxt;
            __builtin_amdgcn_sched_barrier(0);
            // Copy-out: sixteen lanes per node, a lane per piece -- every store instruction writes whole runs of K (four
            // nodes' 32 cnt contiguous bytes each) instead of one 16-byte piece per triangle scattered over the rows.
            // Measured on the 1M mesh with the piecewise stores: the kernel took the same 48 us with the arithmetic
            // removed and 36 us with the stores removed -- 3 M write requests of 37 bytes on average for 111 MB.
            bool longrow = false;
#pragma unroll
            for (int j = 0; j < (NPW + 3) / 4; ++j) {
                const int node = 4 * j + (wl >> 4), piece = wl & 15;
                const int2 h = node < NPW ? hdr[node] : make_int2(0, 0);
                if (piece < 2 * h.y) k_store(kval + 4 * (int64_t)h.x + 2 * piece, wreg[node * kFanStage + piece]);
                longrow |= h.y > 8;
            }
            if (__any(longrow ? 1 : 0)) { // rows of nine or ten blocks (open fans of seven or eight triangles): pieces 16-19
#pragma unroll
                for (int j = 0; j < (NPW + 3) / 4; ++j) {
                    const int node = 4 * j + (wl >> 4), piece = 16 + (wl & 15);
                    const int2 h = node < NPW ? hdr[node] : make_int2(0, 0);
                    if (piece < 2 * h.y) k_store(kval + 4 * (int64_t)h.x + 2 * piece, wreg[node * kFanStage + piece]);
                }
            }
            if (!rowfan && cnt > 0 && k == 0 && i >= 0) {
                s_big[l] = i;
                any_big = true;
            }
        }
    }; // run
    if (td > 0 && llast >= lfirst) {
        if (td <= 6 && six)
            run(std::integral_constant<int, 6>{});
        else
            run(std::integral_constant<int, 8>{});
    } else if (llast >= lfirst) { // a tile without a single triangle: whatever rows its nodes have go to the workgroup
        for (int32_t l = lfirst + lane; l <= llast; l += kFanThreads) {
            const int32_t i = s_cid[l];
            if (bptr[i + 1] - bptr[i] > 0) {
                s_big[l] = i;
                any_big = true;
            }
        }
    }
    // rows the fan scheme did not take: the whole workgroup, one thread per block, k_assemble_rows arithmetic
    if (!__syncthreads_or(any_big ? 1 : 0)) return;
    for (int m = lfirst; m < lend; ++m) {
        const int32_t ib = s_big[m];
        if (ib < 0) continue; // uniform
        const int64_t gb_ = (int64_t)t * B + m;
        const int32_t pb = bptr[ib], cb = bptr[ib + 1] - pb;
        for (int kpos = lane; kpos < cb; kpos += kFanThreads) {
            const int32_t j = bcol[pb + kpos];
            double k00 = 0.0, k01 = 0.0, k10 = 0.0, k11 = 0.0;
            for (int32_t q = inc_off[gb_]; q < inc_off[gb_ + 1]; ++q) {
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
            double *r0 = kval + 4 * (int64_t)pb + 2 * kpos;
            double *r1 = kval + 4 * (int64_t)pb + 2 * cb + 2 * kpos;
            r0[0] = k00;
            r0[1] = k01;
            r1[0] = k10;
            r1[1] = k11;
        }
    }
}

static int32_t asm_img_pieces(int32_t cap, int32_t B) { return cap + (B * 4 + 15) / 16; } // 16-byte pieces of the staged image

size_t assemble_ctiles_lds(int32_t B, int32_t cap)
{
    return (size_t)asm_img_pieces(cap, B) * 16 + (size_t)kFanThreads * 64 + (size_t)B * 4;
}

bool assemble_ctiles(const int32_t *bcol, const int32_t *bptr, const uint32_t *perm, const double *xyP,
                     const double *halo_xy, const int32_t *tile_hoff, const int32_t *tile_deg, const int64_t *tile_off,
                     const uint32_t *ell_asm, const uint16_t *ell_pos, const int32_t *inc_off, const uint32_t *inc,
                     const int32_t *conn, const double *xy, int64_t N, int32_t B, int32_t T, int32_t cap, double nu,
                     double youngs, double thick, double *kval, hipStream_t s)
{
    const size_t lds = assemble_ctiles_lds(B, cap);
    if ((B != 256 && B != 512) || cap > 4096 || lds > 64 * 1024) return false; // 12-bit local ids; the image fits the LDS
    const int32_t img = asm_img_pieces(cap, B);
    // workgroups per tile: at least ~3 per CU on small meshes (measured at 982 tiles: 1 per tile 44.6 us, 2: 49.0; at
    // 3911 tiles 1: 152, 2: 165)
    int32_t segs = 1;
    while ((int64_t)T * segs < 768 && B / (2 * segs) >= kFanThreads / kFanLanes) segs *= 2;
    const char *e6 = getenv("MAG_TUNE_ASM_LANES"); // 8: never six lanes per node
    const int32_t six = e6 && atoi(e6) == 8 ? 0 : 1;
    if (const char *e = getenv("MAG_TUNE_ASM_SEGS")) segs = std::max(1, std::min(atoi(e), B / (kFanThreads / kFanLanes)));
    if (B == 256)
        k_assemble_fan<256><<<T * segs, kFanThreads, lds, s>>>(
            bcol, bptr, perm, (const double2 *)xyP, (const double2 *)halo_xy, tile_hoff, tile_deg, tile_off, ell_asm,
            ell_pos, inc_off, inc, conn, (const double2 *)xy, N, cap, img, segs, six, nu, youngs, thick, kval);
    else
        k_assemble_fan<512><<<T * segs, kFanThreads, lds, s>>>(
            bcol, bptr, perm, (const double2 *)xyP, (const double2 *)halo_xy, tile_hoff, tile_deg, tile_off, ell_asm,
            ell_pos, inc_off, inc, conn, (const double2 *)xy, N, cap, img, segs, six, nu, youngs, thick, kval);
    return true;
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

// Rows with a prescribed column: marked from the PRESCRIBED side.  The pattern is symmetric node by node (solver.rs:304-322:
// every (node, node) pair of an element, both ways) and every rank keeps the rows of the prescribed nodes, so the columns of
// a prescribed node's row are exactly the rows that have it as a column -- O(boundary) work instead of a gather of u_known
// behind every one of the 14 M column entries.
__global__ void __launch_bounds__(256) k_mark_bc_rows(const int32_t *bptr, const int32_t *bcol, const uint8_t *u_known,
                                                      int64_t N, uint8_t *touch)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    if (!(u_known[2 * i] | u_known[2 * i + 1])) return;
    for (int32_t k = bptr[i]; k < bptr[i + 1]; ++k) touch[bcol[k]] = 1; // (racing stores of the same value)
}

// One thread per node of the HILBERT order (b is written in that order: one coalesced 16-byte store per node; the caller-order
// inputs are gathered -- scattered 8-byte stores through the permutation cost more than the whole row sums).
__global__ void __launch_bounds__(256) k_rhs_from_csr(const int32_t *bptr, const int32_t *bcol, const double *kval,
                                                      const uint8_t *u_known, const double *u_in, const double *f_in,
                                                      const uint32_t *perm, const uint8_t *touch, bool hilbert_flags,
                                                      int64_t N, double2 *bP)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= N) return;
    const int64_t i = perm[g];
    const bool t = touch[hilbert_flags ? g : i] != 0;
    double2 v;
    // a row without a prescribed column: rhs_row's sum stays at its 0.0, i.e. 0.0 + f (the same bits, -0.0 included)
    v.x = u_known[2 * i] ? 0.0 : (t ? rhs_row(bptr, bcol, kval, u_known, u_in, f_in, 2 * i) : 0.0 + f_in[2 * i]);
    v.y = u_known[2 * i + 1] ? 0.0 : (t ? rhs_row(bptr, bcol, kval, u_known, u_in, f_in, 2 * i + 1) : 0.0 + f_in[2 * i + 1]);
    bP[g] = v;
}

// (round 4) The right-hand side of a row WITHOUT a prescribed column is b = 0.0 + f (solver.rs:427-432 with an empty known
// part; 0 on a prescribed DOF): the ordering phase writes that for every node while it gathers the node's other data by the
// permutation (symbolic.hip, k_apply_order).  Only the rows that have a prescribed column need K: the pattern kernel flags
// them in Hilbert order (one coalesced byte per thread here; O(sqrt N) rows do the work, the others leave at once).  Round 3's
// kernel gathered flag, mask and forces of EVERY node by caller id: 17.6 us at 1M triangles, three times the row sums.
__global__ void __launch_bounds__(256) k_rhs_touched(const int32_t *bptr, const int32_t *bcol, const double *kval,
                                                     const uint8_t *u_known, const double *u_in, const double *f_in,
                                                     const uint32_t *perm, const uint8_t *touch, int64_t N, double2 *bP)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= N || !touch[g]) return;
    const int64_t i = perm[g];
    // Both rows of the node, eight block columns at a time: the columns, then their flags, then the values behind the
    // prescribed ones are each fetched together (the few threads that work here are latency chains: with rhs_row's one
    // column at a time the kernel took 14 us at 1M triangles for ~1500 rows); the sums run in rhs_row's order -- ascending
    // column, x then y, every term (K u) * -1.0 -- and end with + f.
    const int32_t p = bptr[i], nb = bptr[i + 1] - p;
    const double *row0 = kval + 4 * (int64_t)p, *row1 = row0 + 2 * (int64_t)nb;
    double s0 = 0.0, s1 = 0.0;
    for (int32_t k0 = 0; k0 < nb; k0 += 8) {
        int32_t col[8];
        uint32_t uk[8];
        double2 u[8], a0[8], a1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) col[k] = k0 + k < nb ? bcol[p + k0 + k] : -1;
#pragma unroll
        for (int k = 0; k < 8; ++k) uk[k] = col[k] >= 0 ? ((const uint16_t *)u_known)[col[k]] : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (uk[k]) {
                u[k] = ((const double2 *)u_in)[col[k]];
                a0[k] = ((const double2 *)row0)[k0 + k];
                a1[k] = ((const double2 *)row1)[k0 + k];
            }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (uk[k] & 0xffu) {
                s0 += (a0[k].x * u[k].x) * -1.0;
                s1 += (a1[k].x * u[k].x) * -1.0;
            }
            if (uk[k] >> 8) {
                s0 += (a0[k].y * u[k].y) * -1.0;
                s1 += (a1[k].y * u[k].y) * -1.0;
            }
        }
    }
    const uint32_t own = ((const uint16_t *)u_known)[i];
    const double2 f = ((const double2 *)f_in)[i];
    bP[g] = make_double2((own & 0xffu) ? 0.0 : s0 + f.x, (own >> 8) ? 0.0 : s1 + f.y);
}

void rhs_touched(const int32_t *bptr, const int32_t *bcol, const double *kval, const uint8_t *u_known, const double *u_in,
                 const double *f_in, const uint32_t *perm, const uint8_t *touch, int64_t N, double *bP, hipStream_t s)
{
    k_rhs_touched<<<blocks_for(N, 256), 256, 0, s>>>(bptr, bcol, kval, u_known, u_in, f_in, perm, touch, N, (double2 *)bP);
}

// touch: N bytes; touch_ready: already filled by the pattern kernel (symbolic.hip, k_pattern_rows), else marked here
void rhs_from_csr(const int32_t *bptr, const int32_t *bcol, const double *kval, const uint8_t *u_known,
                  const double *u_in, const double *f_in, const uint32_t *perm, uint8_t *touch, bool touch_ready,
                  int64_t N, double *bP, hipStream_t s)
{
    if (!touch_ready) {
        (void)hipMemsetAsync(touch, 0, (size_t)N, s);
        k_mark_bc_rows<<<blocks_for(N, 256), 256, 0, s>>>(bptr, bcol, u_known, N, touch);
    }
    // (flags from the pattern kernel are in Hilbert order, those marked here in caller numbering)
    k_rhs_from_csr<<<blocks_for(N, 256), 256, 0, s>>>(bptr, bcol, kval, u_known, u_in, f_in, perm, touch, touch_ready, N,
                                                      (double2 *)bP);
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
