/*
 * magnetite_oracle.c -- CPU restatement of Magnetite's solver hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under magnetite_amd/ may include, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (kyle-tennison/Magnetite @ 2024_08_07) holds
 * no tests, golden vectors or fixtures, is a Rust crate (no cargo/rustc in this
 * image) and its CG lives in argmin 0.10.0 / nalgebra-sparse 0.9.0, which are
 * not vendored.  This file follows src/solver.rs line by line (citations on
 * each function) and restates the published algorithms of the third-party
 * crates at the reference's call sites.  It is pinned only by hand-derivable
 * known-answer tests (tests/test_oracle_kat.py), an independent numpy twin and
 * an analytic patch test -- not by reference outputs.
 *
 * Build: gcc -O2 -ffp-contract=off (Rust never contracts a*b+c into an FMA,
 * so neither may this file).
 *
 * Conventions shared with include/magnetite_hip.h:
 *   xy[2N]      node coordinates, interleaved x,y   (datatypes.rs:1-5)
 *   conn[3E]    element node indices                 (datatypes.rs:17-18)
 *   u_known[2N] 1 => displacement prescribed (node.ux/uy is Some, fx/fy None)
 *               0 => force prescribed        (node.fx/fy is Some, ux/uy None)
 *   DOF numbering 2*node + {0:x, 1:y}               (solver.rs:306-307,346-351)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_STOP_RNORM 0    /* cost = sqrt(r.r)  <= tol (absolute)          */
#define ORC_STOP_RNORM_SQ 1 /* cost = r.r        <= tol (absolute)          */
#define ORC_STOP_REL 2      /* sqrt(r.r) <= tol * sqrt(b.b) (not reference) */

typedef struct {
    int64_t n;    /* rows == cols */
    int64_t nnz;
    int64_t *rowptr; /* n+1 */
    int32_t *col;    /* ascending within each row */
    double *val;
} orc_csr;

/* ---------------------------------------------------------------- K_e --- */

/* solver.rs:187-193  compute_element_area -- SIGNED area. */
double orc_element_area(const double *xy, const int32_t *tri)
{
    const double x0 = xy[2 * tri[0]], y0 = xy[2 * tri[0] + 1];
    const double x1 = xy[2 * tri[1]], y1 = xy[2 * tri[1] + 1];
    const double x2 = xy[2 * tri[2]], y2 = xy[2 * tri[2] + 1];
    return 0.5 * (x0 * (y1 - y2) + x1 * (y2 - y0) + x2 * (y0 - y1));
}

/* solver.rs:204-230  compute_strain_displacement_matrix -- B (3x6, row major)
 * entries divided (true division) by 2*area. */
void orc_strain_displacement(const double *xy, const int32_t *tri, double area, double *B)
{
    const double x0 = xy[2 * tri[0]], y0 = xy[2 * tri[0] + 1];
    const double x1 = xy[2 * tri[1]], y1 = xy[2 * tri[1] + 1];
    const double x2 = xy[2 * tri[2]], y2 = xy[2 * tri[2] + 1];
    const double b1 = y1 - y2, b2 = y2 - y0, b3 = y0 - y1;
    const double g1 = x2 - x1, g2 = x0 - x2, g3 = x1 - x0;
    const double m[18] = {b1, 0., b2, 0., b3, 0., 0., g1, 0., g2, 0., g3, g1, b1, g2, b2, g3, b3};
    const double d = 2.0 * area;
    for (int i = 0; i < 18; ++i) B[i] = m[i] / d;
}

/* solver.rs:240-250  compute_stress_strain_matrix -- D (3x3, row major),
 * plane stress, scaled elementwise by E / (1 - nu^2). */
void orc_stress_strain(double nu, double youngs, double *D)
{
    const double m[9] = {1.0, nu, 0.0, nu, 1.0, 0.0, 0.0, 0.0, (1.0 - nu) / 2.0};
    const double s = youngs / (1.0 - nu * nu); /* f64::powi(nu,2) == nu*nu */
    for (int i = 0; i < 9; ++i) D[i] = m[i] * s;
}

/* nalgebra 0.32 static matrix product as the reference uses it
 * (solver.rs:274-275,516-522): gemm -> per output column gemv -> axcpy, i.e.
 * C[i,j] = (((a[i,0]*b[0,j]) + a[i,1]*b[1,j]) + ...) ascending k, no FMA. */
static void matmul(const double *a, const double *b, double *c, int m, int k, int n)
{
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = a[i * k] * b[j];
            for (int l = 1; l < k; ++l) s = s + a[i * k + l] * b[l * n + j];
            c[i * n + j] = s;
        }
}

/* solver.rs:263-278  compute_element_stiffness_matrix
 * K_e = ((B^T * D) * B) * area * thickness, area signed. Row major 6x6. */
void orc_element_stiffness(const double *xy, const int32_t *tri, double nu, double youngs,
                           double thickness, double *Ke)
{
    double B[18], Bt[18], D[9], BtD[18];
    const double area = orc_element_area(xy, tri);
    orc_stress_strain(nu, youngs, D);
    orc_strain_displacement(xy, tri, area, B);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 6; ++j) Bt[j * 3 + i] = B[i * 6 + j];
    matmul(Bt, D, BtD, 6, 3, 3);
    matmul(BtD, B, Ke, 6, 3, 6);
    for (int i = 0; i < 36; ++i) Ke[i] = Ke[i] * area * thickness;
}

/* solver.rs:548-567  the element loop of run(): one K_e per element. */
void orc_element_stiffness_all(int64_t E, const double *xy, const int32_t *conn, double nu,
                               double youngs, double thickness, double *Ke)
{
    for (int64_t e = 0; e < E; ++e)
        orc_element_stiffness(xy, conn + 3 * e, nu, youngs, thickness, Ke + 36 * e);
}

/* ------------------------------------------------------ dense assembly --- */

/* solver.rs:290-331  build_total_stiffness_matrix -- dense n x n, element
 * order, 36 '+=' per element.  K is row major here (layout does not change
 * values). */
void orc_assemble_dense(int64_t N, int64_t E, const int32_t *conn, const double *Ke, double *K)
{
    const int64_t n = 2 * N;
    memset(K, 0, sizeof(double) * (size_t)n * (size_t)n);
    for (int64_t e = 0; e < E; ++e) {
        const double *k = Ke + 36 * e;
        for (int lr = 0; lr < 3; ++lr)
            for (int lc = 0; lc < 3; ++lc) {
                const int64_t gr = 2 * (int64_t)conn[3 * e + lr], gc = 2 * (int64_t)conn[3 * e + lc];
                K[gr * n + gc] += k[(2 * lr) * 6 + 2 * lc];
                K[gr * n + gc + 1] += k[(2 * lr) * 6 + 2 * lc + 1];
                K[(gr + 1) * n + gc] += k[(2 * lr + 1) * 6 + 2 * lc];
                K[(gr + 1) * n + gc + 1] += k[(2 * lr + 1) * 6 + 2 * lc + 1];
            }
    }
}

/* solver.rs:365-404 + 427-432: rows = DOFs with known force (ascending),
 * unknown = K[row, unknown-u cols]; known[r,k] = -(K[row,col]*u_known[col]);
 * b = column_sum(known) (ascending known index, starting from 0) + f_known.
 * Kff is (nf x nf) row major, b length nf. */
void orc_partition_dense(int64_t n, const double *K, const uint8_t *u_known, const double *u_in,
                         const double *f_in, double *Kff, double *b)
{
    int64_t nf = 0;
    for (int64_t i = 0; i < n; ++i) nf += !u_known[i];
    int64_t lr = 0;
    for (int64_t row = 0; row < n; ++row) {
        if (u_known[row]) continue; /* nodal_force.is_none() */
        int64_t uc = 0;
        double s = 0.0;
        for (int64_t col = 0; col < n; ++col) {
            if (u_known[col])
                s += (K[row * n + col] * u_in[col]) * -1.0;
            else
                Kff[lr * nf + uc++] = K[row * n + col];
        }
        b[lr] = s + f_in[row];
        ++lr;
    }
}

/* solver.rs:123-137  dense -> COO (row major scan, exact zeros dropped) ->
 * CsrMatrix::from(&coo).  Caller frees with orc_csr_free. */
orc_csr *orc_sparsify_dense(int64_t n, const double *A)
{
    orc_csr *m = (orc_csr *)calloc(1, sizeof(orc_csr));
    m->n = n;
    m->rowptr = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    int64_t nnz = 0;
    for (int64_t i = 0; i < n * n; ++i) nnz += (A[i] != 0.0);
    m->nnz = nnz;
    m->col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
    m->val = (double *)malloc(sizeof(double) * (size_t)(nnz ? nnz : 1));
    int64_t p = 0;
    for (int64_t r = 0; r < n; ++r) {
        m->rowptr[r] = p;
        for (int64_t c = 0; c < n; ++c)
            if (A[r * n + c] != 0.0) {
                m->col[p] = (int32_t)c;
                m->val[p++] = A[r * n + c];
            }
    }
    m->rowptr[n] = p;
    return m;
}

void orc_csr_free(orc_csr *m)
{
    if (!m) return;
    free(m->rowptr);
    free(m->col);
    free(m->val);
    free(m);
}
int64_t orc_csr_n(const orc_csr *m) { return m->n; }
int64_t orc_csr_nnz(const orc_csr *m) { return m->nnz; }
void orc_csr_copy(const orc_csr *m, int64_t *rowptr, int32_t *col, double *val)
{
    memcpy(rowptr, m->rowptr, sizeof(int64_t) * (size_t)(m->n + 1));
    memcpy(col, m->col, sizeof(int32_t) * (size_t)m->nnz);
    memcpy(val, m->val, sizeof(double) * (size_t)m->nnz);
}

/* ------------------------------------------------------------ SpMV, CG --- */

/* solver.rs:31-36  &CsrMatrix * DVector (nalgebra-sparse 0.9 spmm_csr_dense):
 * y zeroed, then y[i] += a_ik * x[k] over the row's ascending columns. */
void orc_spmv(const orc_csr *A, const double *x, double *y)
{
    for (int64_t i = 0; i < A->n; ++i) {
        double s = 0.0;
        for (int64_t p = A->rowptr[i]; p < A->rowptr[i + 1]; ++p) s += A->val[p] * x[A->col[p]];
        y[i] = s;
    }
}

/* argmin-math 0.4 Vec<f64> dot: sequential sum of products from 0.0. */
static double vdot(const double *a, const double *b, int64_t n)
{
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* solver.rs:139-176  argmin 0.10 ConjugateGradient under Executor:
 *   init : r0 = -(b - A x0) (x0 = 0, solver.rs:143), p0 = -r0, rtr = r0.r0
 *   iter : q = A p; alpha = rtr/(p.q); x += alpha p; r += alpha q;
 *          rtr' = r.r; beta = rtr'/rtr; p = -r + beta p; cost = f(rtr')
 *   stop : best_cost <= target_cost (solver.rs:154) or iter >= max_iters (:153)
 *   out  : best_param (solver.rs:167-174) -- the iterate with the lowest cost.
 * Executor order: termination is tested at the top of the loop on the state
 * left by the previous iteration, so the solver stops after the first
 * iteration whose cost is <= target.  `history` (may be NULL) receives the
 * cost of iterations 1..hist_len.  Returns iterations executed. */
int64_t orc_cg(const orc_csr *A, const double *b, int stop_mode, double tol, int64_t max_iter,
               double *x_best, double *final_cost, double *history, int64_t hist_len)
{
    const int64_t n = A->n;
    double *x = (double *)calloc((size_t)n + 1, sizeof(double));
    double *r = (double *)malloc(sizeof(double) * ((size_t)n + 1));
    double *p = (double *)malloc(sizeof(double) * ((size_t)n + 1));
    double *q = (double *)malloc(sizeof(double) * ((size_t)n + 1));
    /* init: A*x0 with x0 = 0 is 0, r0 = (b - 0) * -1, p0 = r0 * -1 */
    orc_spmv(A, x, q);
    for (int64_t i = 0; i < n; ++i) {
        r[i] = (b[i] - q[i]) * -1.0;
        p[i] = r[i] * -1.0;
    }
    double rtr = vdot(r, r, n);
    double target = tol;
    if (stop_mode == ORC_STOP_REL) target = tol * sqrt(vdot(b, b, n));
    double best = INFINITY;
    /* b == 0 would make the first alpha 0/0; what argmin does there cannot be
     * checked offline, so both this oracle and the HIP path return x = 0. */
    if (rtr == 0.0) best = 0.0;
    memcpy(x_best, x, sizeof(double) * (size_t)n);
    int64_t it = 0;
    while (it < max_iter && !(best <= target)) {
        orc_spmv(A, p, q);
        const double alpha = rtr / vdot(p, q, n);
        for (int64_t i = 0; i < n; ++i) x[i] = x[i] + alpha * p[i];
        for (int64_t i = 0; i < n; ++i) r[i] = r[i] + alpha * q[i];
        const double rtr_n = vdot(r, r, n);
        const double beta = rtr_n / rtr;
        rtr = rtr_n;
        for (int64_t i = 0; i < n; ++i) p[i] = r[i] * -1.0 + beta * p[i];
        const double cost = (stop_mode == ORC_STOP_RNORM_SQ) ? fabs(rtr_n) : sqrt(rtr_n);
        if (history && it < hist_len) history[it] = cost;
        ++it;
        if (cost < best) {
            best = cost;
            memcpy(x_best, x, sizeof(double) * (size_t)n);
        }
    }
    if (final_cost) *final_cost = best;
    free(x);
    free(r);
    free(p);
    free(q);
    return it;
}

/* All-cores variant of the same CG ("also report an OpenMP variant on all host cores"): rows of the SpMV
 * and the vector updates are split over threads, dot products are OpenMP reductions (summation order differs from
 * argmin's sequential sums).  The reference itself is single-threaded; this exists only so the CPU baseline is not
 * flattered by idle cores.  Returns iterations executed; x holds the last iterate. */
int64_t orc_cg_parallel(const orc_csr *A, const double *b, int stop_mode, double tol, int64_t max_iter, double *x,
                        double *final_cost)
{
    const int64_t n = A->n;
    double *r = (double *)malloc(sizeof(double) * ((size_t)n + 1));
    double *p = (double *)malloc(sizeof(double) * ((size_t)n + 1));
    double *q = (double *)malloc(sizeof(double) * ((size_t)n + 1));
    double rtr = 0.0, bb = 0.0;
#pragma omp parallel for reduction(+ : rtr, bb) schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        x[i] = 0.0;
        r[i] = -b[i];
        p[i] = b[i];
        rtr += b[i] * b[i];
        bb += b[i] * b[i];
    }
    const double target = stop_mode == ORC_STOP_REL ? tol * sqrt(bb) : tol;
    double cost = INFINITY;
    int64_t it = 0;
    if (rtr == 0.0) cost = 0.0;
    while (it < max_iter && !(cost <= target)) {
        double pq = 0.0;
#pragma omp parallel for reduction(+ : pq) schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            double s = 0.0;
            for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k) s += A->val[k] * p[A->col[k]];
            q[i] = s;
            pq += p[i] * s;
        }
        const double alpha = rtr / pq;
        double rtr_n = 0.0;
#pragma omp parallel for reduction(+ : rtr_n) schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            x[i] = x[i] + alpha * p[i];
            r[i] = r[i] + alpha * q[i];
            rtr_n += r[i] * r[i];
        }
        const double beta = rtr_n / rtr;
        rtr = rtr_n;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) p[i] = r[i] * -1.0 + beta * p[i];
        cost = (stop_mode == ORC_STOP_RNORM_SQ) ? fabs(rtr_n) : sqrt(rtr_n);
        ++it;
    }
    if (final_cost) *final_cost = cost;
    free(r);
    free(p);
    free(q);
    return it;
}

/* ------------------------------------- opt-in preconditioning (no reference) --- */

/* SURVEY 8f rank 4.  The reference runs plain CG (solver.rs:142-157); this is an ADDITION with no reference
 * counterpart, off by default on the HIP path, restated here only so the HIP kernels have a checker.
 * M = the 2x2 node-diagonal blocks of K_ff (kind 2, block-Jacobi) or its diagonal (kind 1, Jacobi).  The inverse
 * blocks are computed in double, then ROUNDED TO FLOAT (the HIP path stores them in fp32: any fixed SPD M gives
 * the same solution, and 16 bytes per node is cheaper to stream than 32).  minv holds (i00, i01, i11) per node;
 * a prescribed DOF drops out of its block (its residual is identically 0). */
void orc_block_jacobi(const orc_csr *K, int64_t N, const uint8_t *u_known, int kind, float *minv)
{
    for (int64_t i = 0; i < N; ++i) {
        double k00 = 0.0, k01 = 0.0, k11 = 0.0;
        for (int64_t p = K->rowptr[2 * i]; p < K->rowptr[2 * i + 1]; ++p) {
            if (K->col[p] == 2 * i) k00 = K->val[p];
            if (K->col[p] == 2 * i + 1) k01 = K->val[p];
        }
        for (int64_t p = K->rowptr[2 * i + 1]; p < K->rowptr[2 * i + 2]; ++p)
            if (K->col[p] == 2 * i + 1) k11 = K->val[p];
        const int fx = !u_known[2 * i], fy = !u_known[2 * i + 1];
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
        minv[3 * i] = (float)i00;
        minv[3 * i + 1] = (float)i01;
        minv[3 * i + 2] = (float)i11;
    }
}

/* Preconditioned CG in argmin's sign convention (r = A x - b, p = -z + beta p), textbook recurrences:
 * alpha = r.z / p.Ap, beta = r'.z' / r.z.  Stop rule, cost (the TRUE residual norm, not the M-norm), best-iterate
 * and b == 0 handling as orc_cg.  ms/mc/partner describe z = M^-1 r in the reduced numbering:
 * z[f] = ms[f] * r[f] + (partner[f] >= 0 ? mc[f] * r[partner[f]] : 0). */
int64_t orc_pcg(const orc_csr *A, const double *b, const double *ms, const double *mc, const int64_t *partner,
                int stop_mode, double tol, int64_t max_iter, double *x_best, double *final_cost, double *history,
                int64_t hist_len)
{
    const int64_t n = A->n;
    double *x = (double *)calloc((size_t)n + 1, sizeof(double));
    double *r = (double *)malloc(sizeof(double) * ((size_t)n + 1));
    double *z = (double *)malloc(sizeof(double) * ((size_t)n + 1));
    double *p = (double *)malloc(sizeof(double) * ((size_t)n + 1));
    double *q = (double *)malloc(sizeof(double) * ((size_t)n + 1));
    for (int64_t i = 0; i < n; ++i) r[i] = b[i] * -1.0;
    for (int64_t i = 0; i < n; ++i) {
        z[i] = ms[i] * r[i] + (partner[i] >= 0 ? mc[i] * r[partner[i]] : 0.0);
        p[i] = z[i] * -1.0;
    }
    double rho = vdot(r, z, n);
    const double rtr0 = vdot(r, r, n);
    double target = tol;
    if (stop_mode == ORC_STOP_REL) target = tol * sqrt(vdot(b, b, n));
    double best = INFINITY;
    if (rtr0 == 0.0) best = 0.0;
    memcpy(x_best, x, sizeof(double) * (size_t)n);
    int64_t it = 0;
    while (it < max_iter && !(best <= target)) {
        orc_spmv(A, p, q);
        const double alpha = rho / vdot(p, q, n);
        for (int64_t i = 0; i < n; ++i) x[i] = x[i] + alpha * p[i];
        for (int64_t i = 0; i < n; ++i) r[i] = r[i] + alpha * q[i];
        for (int64_t i = 0; i < n; ++i) z[i] = ms[i] * r[i] + (partner[i] >= 0 ? mc[i] * r[partner[i]] : 0.0);
        const double rho_n = vdot(r, z, n);
        const double beta = rho_n / rho;
        rho = rho_n;
        for (int64_t i = 0; i < n; ++i) p[i] = z[i] * -1.0 + beta * p[i];
        const double rtr = vdot(r, r, n);
        const double cost = (stop_mode == ORC_STOP_RNORM_SQ) ? fabs(rtr) : sqrt(rtr);
        if (history && it < hist_len) history[it] = cost;
        ++it;
        if (cost < best) {
            best = cost;
            memcpy(x_best, x, sizeof(double) * (size_t)n);
        }
    }
    if (final_cost) *final_cost = best;
    free(x);
    free(r);
    free(z);
    free(p);
    free(q);
    return it;
}

/* --------------------------------------------------------- post-solve --- */

/* solver.rs:496-535  compute_stress: sigma = (D*B)*u_e, scalar =
 * sqrt(sx^2+sy^2) * (sx+sy < 1.0 ? -1 : 1)  (the `< 1.0` quirk is kept). */
void orc_stress(int64_t E, const double *xy, const int32_t *conn, const double *u, double nu,
                double youngs, double *stress)
{
    for (int64_t e = 0; e < E; ++e) {
        const int32_t *tri = conn + 3 * e;
        double D[9], B[18], DB[18], ue[6], s[3];
        for (int c = 0; c < 3; ++c) {
            ue[2 * c] = u[2 * tri[c]];
            ue[2 * c + 1] = u[2 * tri[c] + 1];
        }
        orc_stress_strain(nu, youngs, D);
        orc_strain_displacement(xy, tri, orc_element_area(xy, tri), B);
        matmul(D, B, DB, 3, 3, 6);
        matmul(DB, ue, s, 3, 6, 1);
        const double sign = (s[0] + s[1] < 1.0) ? -1.0 : 1.0;
        stress[e] = sqrt(s[0] * s[0] + s[1] * s[1]) * sign;
    }
}

/* ------------------------------------------- reference-faithful dense --- */

typedef struct {
    int64_t iterations;
    double final_cost;
    int64_t n_free;
    int64_t nnz_ff;
} orc_stats;

/* solver.rs:543-586 run() with every O(n^2) intermediate, as written.
 * Small meshes only (2N <= ~20k). Returns 0, or -1 if the BC set leaves no
 * unknown. */
int orc_run_dense(int64_t N, int64_t E, const double *xy, const int32_t *conn,
                  const uint8_t *u_known, const double *u_in, const double *f_in, double youngs,
                  double nu, double thickness, int stop_mode, double tol, int64_t max_iter,
                  double *u_out, double *f_out, double *stress_out, orc_stats *st,
                  double *history, int64_t hist_len)
{
    const int64_t n = 2 * N;
    double *Ke = (double *)malloc(sizeof(double) * 36 * (size_t)(E ? E : 1));
    orc_element_stiffness_all(E, xy, conn, nu, youngs, thickness, Ke);
    double *K = (double *)malloc(sizeof(double) * (size_t)n * (size_t)n);
    orc_assemble_dense(N, E, conn, Ke, K);
    free(Ke);
    int64_t nf = 0;
    for (int64_t i = 0; i < n; ++i) nf += !u_known[i];
    double *Kff = (double *)calloc((size_t)(nf * nf) + 1, sizeof(double));
    double *b = (double *)calloc((size_t)nf + 1, sizeof(double));
    orc_partition_dense(n, K, u_known, u_in, f_in, Kff, b);
    orc_csr *A = orc_sparsify_dense(nf, Kff);
    free(Kff);
    double *xs = (double *)calloc((size_t)nf + 1, sizeof(double));
    double cost = 0.0;
    const int64_t it = orc_cg(A, b, stop_mode, tol, max_iter, xs, &cost, history, hist_len);
    if (st) {
        st->iterations = it;
        st->final_cost = cost;
        st->n_free = nf;
        st->nnz_ff = A->nnz;
    }
    orc_csr_free(A);
    free(b);
    /* solver.rs:443-454 scatter-back, ascending DOF */
    int64_t cur = 0;
    for (int64_t i = 0; i < n; ++i) u_out[i] = u_known[i] ? u_in[i] : xs[cur++];
    free(xs);
    /* solver.rs:456-469 reactions: full dense row . u, ascending column */
    for (int64_t i = 0; i < n; ++i) {
        if (!u_known[i]) {
            f_out[i] = f_in[i];
            continue;
        }
        double s = 0.0;
        for (int64_t c = 0; c < n; ++c) s += K[i * n + c] * u_out[c];
        f_out[i] = s;
    }
    free(K);
    orc_stress(E, xy, conn, u_out, nu, youngs, stress_out);
    return 0;
}

/* -------------------------------------------------- sparse restatement --- */
/* Same arithmetic, same summation orders, no O(n^2) storage: K is built
 * directly in CSR on the node-adjacency pattern; every '+=' of
 * solver.rs:312-322 lands in element order on the same entry, so values are
 * bit-identical to the dense path (tests/test_oracle_paths.py pins that). */

static int cmp_i64(const void *a, const void *b)
{
    const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

/* full K (2N x 2N) on the structural pattern (explicit zeros kept: the dense
 * K holds them too). */
orc_csr *orc_assemble_sparse(int64_t N, int64_t E, const double *xy, const int32_t *conn, double nu,
                             double youngs, double thickness)
{
    /* node adjacency: unique sorted (i,j) node pairs over all elements */
    const int64_t np = 9 * E;
    int64_t *pairs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(np ? np : 1));
    for (int64_t e = 0; e < E; ++e)
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                pairs[9 * e + 3 * a + b] = ((int64_t)conn[3 * e + a] << 32) | (int64_t)conn[3 * e + b];
    qsort(pairs, (size_t)np, sizeof(int64_t), cmp_i64);
    int64_t nb = 0;
    for (int64_t i = 0; i < np; ++i)
        if (i == 0 || pairs[i] != pairs[i - 1]) pairs[nb++] = pairs[i];
    int64_t *bptr = (int64_t *)calloc((size_t)N + 2, sizeof(int64_t));
    for (int64_t i = 0; i < nb; ++i) bptr[(pairs[i] >> 32) + 1]++;
    for (int64_t i = 0; i < N; ++i) bptr[i + 1] += bptr[i];
    orc_csr *m = (orc_csr *)calloc(1, sizeof(orc_csr));
    m->n = 2 * N;
    m->nnz = 4 * nb;
    m->rowptr = (int64_t *)calloc((size_t)m->n + 1, sizeof(int64_t));
    m->col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m->nnz ? m->nnz : 1));
    m->val = (double *)calloc((size_t)(m->nnz ? m->nnz : 1), sizeof(double));
    for (int64_t i = 0; i < N; ++i) {
        const int64_t cnt = bptr[i + 1] - bptr[i];
        m->rowptr[2 * i] = 4 * bptr[i];
        m->rowptr[2 * i + 1] = 4 * bptr[i] + 2 * cnt;
        for (int64_t k = 0; k < cnt; ++k) {
            const int32_t j = (int32_t)(pairs[bptr[i] + k] & 0xffffffff);
            m->col[4 * bptr[i] + 2 * k] = 2 * j;
            m->col[4 * bptr[i] + 2 * k + 1] = 2 * j + 1;
            m->col[4 * bptr[i] + 2 * cnt + 2 * k] = 2 * j;
            m->col[4 * bptr[i] + 2 * cnt + 2 * k + 1] = 2 * j + 1;
        }
    }
    m->rowptr[m->n] = m->nnz;
    /* scatter in element order (solver.rs:299-325) */
    for (int64_t e = 0; e < E; ++e) {
        double k[36];
        orc_element_stiffness(xy, conn + 3 * e, nu, youngs, thickness, k);
        for (int lr = 0; lr < 3; ++lr)
            for (int lc = 0; lc < 3; ++lc) {
                const int64_t i = conn[3 * e + lr], j = conn[3 * e + lc];
                /* locate block (i,j) */
                int64_t lo = bptr[i], hi = bptr[i + 1] - 1;
                while (lo < hi) {
                    const int64_t mid = (lo + hi) / 2;
                    if ((pairs[mid] & 0xffffffff) < j)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                const int64_t kpos = lo - bptr[i], cnt = bptr[i + 1] - bptr[i];
                double *r0 = m->val + 4 * bptr[i] + 2 * kpos;
                double *r1 = m->val + 4 * bptr[i] + 2 * cnt + 2 * kpos;
                r0[0] += k[(2 * lr) * 6 + 2 * lc];
                r0[1] += k[(2 * lr) * 6 + 2 * lc + 1];
                r1[0] += k[(2 * lr + 1) * 6 + 2 * lc];
                r1[1] += k[(2 * lr + 1) * 6 + 2 * lc + 1];
            }
    }
    free(pairs);
    free(bptr);
    return m;
}

/* All-cores variant of orc_assemble_sparse (bench.py's cpu_baseline.all_cores leg).
 * Rows of K are independent: the reference couples them only through shared elements (solver.rs:304-322), so a
 * thread owns a range of row NODES and, per node, walks the node's incident elements in ascending element order --
 * the order in which solver.rs:299-325 adds into that row -- adding rows 2a, 2a+1 of K_e (a = the node's corner in
 * the element).  Same pattern, same '+=' order per entry: bit-identical to the serial path
 * (tests/test_oracle_paths.py).  An element is evaluated once per corner (3x the flops of the serial loop). */
static int cmp_i32(const void *a, const void *b)
{
    const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

orc_csr *orc_assemble_sparse_omp(int64_t N, int64_t E, const double *xy, const int32_t *conn, double nu,
                                 double youngs, double thickness)
{
    /* incidence lists: (element, corner) per node, ascending element (counting sort, serial O(E): a few ms) */
    int64_t *iptr = (int64_t *)calloc((size_t)N + 2, sizeof(int64_t));
    for (int64_t k = 0; k < 3 * E; ++k) iptr[conn[k] + 1]++;
    for (int64_t i = 0; i < N; ++i) iptr[i + 1] += iptr[i];
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * ((size_t)N + 1));
    memcpy(fill, iptr, sizeof(int64_t) * ((size_t)N + 1));
    int64_t *inc = (int64_t *)malloc(sizeof(int64_t) * (size_t)(3 * E + 1));
    for (int64_t k = 0; k < 3 * E; ++k) inc[fill[conn[k]]++] = k; /* k = 3e + corner, ascending in e per node */
    free(fill);
    /* pattern: distinct nodes of the row node's incident elements, ascending */
    int64_t *bptr = (int64_t *)calloc((size_t)N + 2, sizeof(int64_t));
#pragma omp parallel
    {
        int32_t *tmp = NULL;
        int64_t cap = 0;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < N; ++i) {
            const int64_t d = iptr[i + 1] - iptr[i];
            if (3 * d > cap) { cap = 3 * d + 64; tmp = (int32_t *)realloc(tmp, sizeof(int32_t) * (size_t)cap); }
            int64_t m = 0;
            for (int64_t q = iptr[i]; q < iptr[i + 1]; ++q) {
                const int64_t e = inc[q] / 3;
                for (int c = 0; c < 3; ++c) tmp[m++] = conn[3 * e + c];
            }
            qsort(tmp, (size_t)m, sizeof(int32_t), cmp_i32);
            int64_t u = 0;
            for (int64_t q = 0; q < m; ++q)
                if (q == 0 || tmp[q] != tmp[q - 1]) u++;
            bptr[i + 1] = u;
        }
        free(tmp);
    }
    for (int64_t i = 0; i < N; ++i) bptr[i + 1] += bptr[i];
    const int64_t nb = bptr[N];
    orc_csr *m = (orc_csr *)calloc(1, sizeof(orc_csr));
    m->n = 2 * N;
    m->nnz = 4 * nb;
    m->rowptr = (int64_t *)calloc((size_t)m->n + 1, sizeof(int64_t));
    m->col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m->nnz ? m->nnz : 1));
    m->val = (double *)malloc(sizeof(double) * (size_t)(m->nnz ? m->nnz : 1));
    m->rowptr[m->n] = m->nnz;
#pragma omp parallel
    {
        int32_t *tmp = NULL;
        int64_t cap = 0;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < N; ++i) {
            const int64_t d = iptr[i + 1] - iptr[i], cnt = bptr[i + 1] - bptr[i];
            if (3 * d > cap) { cap = 3 * d + 64; tmp = (int32_t *)realloc(tmp, sizeof(int32_t) * (size_t)cap); }
            int64_t mm = 0;
            for (int64_t q = iptr[i]; q < iptr[i + 1]; ++q) {
                const int64_t e = inc[q] / 3;
                for (int c = 0; c < 3; ++c) tmp[mm++] = conn[3 * e + c];
            }
            qsort(tmp, (size_t)mm, sizeof(int32_t), cmp_i32);
            int64_t u = 0;
            for (int64_t q = 0; q < mm; ++q)
                if (q == 0 || tmp[q] != tmp[q - 1]) tmp[u++] = tmp[q];
            m->rowptr[2 * i] = 4 * bptr[i];
            m->rowptr[2 * i + 1] = 4 * bptr[i] + 2 * cnt;
            int32_t *c0 = m->col + 4 * bptr[i], *c1 = c0 + 2 * cnt;
            double *r0 = m->val + 4 * bptr[i], *r1 = r0 + 2 * cnt;
            for (int64_t k = 0; k < cnt; ++k) {
                c0[2 * k] = c1[2 * k] = 2 * tmp[k];
                c0[2 * k + 1] = c1[2 * k + 1] = 2 * tmp[k] + 1;
                r0[2 * k] = r0[2 * k + 1] = r1[2 * k] = r1[2 * k + 1] = 0.0;
            }
            for (int64_t q = iptr[i]; q < iptr[i + 1]; ++q) { /* ascending element order: solver.rs:299-325 */
                const int64_t e = inc[q] / 3;
                const int lr = (int)(inc[q] % 3);
                double k[36];
                orc_element_stiffness(xy, conn + 3 * e, nu, youngs, thickness, k);
                for (int lc = 0; lc < 3; ++lc) {
                    const int32_t j = conn[3 * e + lc];
                    int64_t lo = 0, hi = cnt - 1;
                    while (lo < hi) {
                        const int64_t mid = (lo + hi) / 2;
                        if (tmp[mid] < j) lo = mid + 1; else hi = mid;
                    }
                    r0[2 * lo] += k[(2 * lr) * 6 + 2 * lc];
                    r0[2 * lo + 1] += k[(2 * lr) * 6 + 2 * lc + 1];
                    r1[2 * lo] += k[(2 * lr + 1) * 6 + 2 * lc];
                    r1[2 * lo + 1] += k[(2 * lr + 1) * 6 + 2 * lc + 1];
                }
            }
        }
        free(tmp);
    }
    free(inc);
    free(iptr);
    free(bptr);
    return m;
}

/* All-cores variant of orc_reduce_system: rows are independent; count -> prefix sum -> fill. */
orc_csr *orc_reduce_system_omp(const orc_csr *K, const uint8_t *u_known, const double *u_in,
                               const double *f_in, double *b /* n_free */)
{
    const int64_t n = K->n;
    int32_t *fidx = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    int64_t nf = 0;
    for (int64_t i = 0; i < n; ++i) fidx[i] = u_known[i] ? -1 : (int32_t)nf++;
    orc_csr *m = (orc_csr *)calloc(1, sizeof(orc_csr));
    m->n = nf;
    m->rowptr = (int64_t *)calloc((size_t)nf + 2, sizeof(int64_t));
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        if (u_known[r]) continue;
        int64_t c = 0;
        for (int64_t p = K->rowptr[r]; p < K->rowptr[r + 1]; ++p)
            c += (!u_known[K->col[p]] && K->val[p] != 0.0);
        m->rowptr[fidx[r] + 1] = c;
    }
    for (int64_t i = 0; i < nf; ++i) m->rowptr[i + 1] += m->rowptr[i];
    const int64_t nnz = m->rowptr[nf];
    m->nnz = nnz;
    m->col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
    m->val = (double *)malloc(sizeof(double) * (size_t)(nnz ? nnz : 1));
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        if (u_known[r]) continue;
        const int64_t lr = fidx[r];
        int64_t q = m->rowptr[lr];
        double s = 0.0;
        for (int64_t p = K->rowptr[r]; p < K->rowptr[r + 1]; ++p) {
            const int32_t c = K->col[p];
            if (u_known[c])
                s += (K->val[p] * u_in[c]) * -1.0;
            else if (K->val[p] != 0.0) {
                m->col[q] = fidx[c];
                m->val[q++] = K->val[p];
            }
        }
        b[lr] = s + f_in[r];
    }
    free(fidx);
    return m;
}

/* solver.rs:365-404,427-432,123-137 on the CSR of K: K_ff in compact unknown
 * numbering with exact zeros dropped, and b = sum_known -(K*u) + f (ascending
 * column; skipped structural zeros contribute exact +-0). */
orc_csr *orc_reduce_system(const orc_csr *K, const uint8_t *u_known, const double *u_in,
                           const double *f_in, double *b /* n_free */)
{
    const int64_t n = K->n;
    int32_t *fidx = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
    int64_t nf = 0;
    for (int64_t i = 0; i < n; ++i) fidx[i] = u_known[i] ? -1 : (int32_t)nf++;
    orc_csr *m = (orc_csr *)calloc(1, sizeof(orc_csr));
    m->n = nf;
    m->rowptr = (int64_t *)calloc((size_t)nf + 1, sizeof(int64_t));
    int64_t nnz = 0;
    for (int64_t r = 0; r < n; ++r) {
        if (u_known[r]) continue;
        for (int64_t p = K->rowptr[r]; p < K->rowptr[r + 1]; ++p)
            nnz += (!u_known[K->col[p]] && K->val[p] != 0.0);
    }
    m->nnz = nnz;
    m->col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
    m->val = (double *)malloc(sizeof(double) * (size_t)(nnz ? nnz : 1));
    int64_t q = 0;
    for (int64_t r = 0; r < n; ++r) {
        if (u_known[r]) continue;
        const int64_t lr = fidx[r];
        m->rowptr[lr] = q;
        double s = 0.0;
        for (int64_t p = K->rowptr[r]; p < K->rowptr[r + 1]; ++p) {
            const int32_t c = K->col[p];
            if (u_known[c])
                s += (K->val[p] * u_in[c]) * -1.0;
            else if (K->val[p] != 0.0) {
                m->col[q] = fidx[c];
                m->val[q++] = K->val[p];
            }
        }
        b[lr] = s + f_in[r];
    }
    m->rowptr[nf] = q;
    free(fidx);
    return m;
}

/* run() on the sparse restatement.  cg_iter_cap > 0 stops CG after that many
 * iterations regardless of cost (bench.py's bounded cpu_baseline sample). */
static int run_sparse_impl(int64_t N, int64_t E, const double *xy, const int32_t *conn,
                           const uint8_t *u_known, const double *u_in, const double *f_in, double youngs,
                           double nu, double thickness, int stop_mode, double tol, int64_t max_iter,
                           double *u_out, double *f_out, double *stress_out, orc_stats *st,
                           double *history, int64_t hist_len, int precond)
{
    const int64_t n = 2 * N;
    orc_csr *K = orc_assemble_sparse(N, E, xy, conn, nu, youngs, thickness);
    int64_t nf = 0;
    for (int64_t i = 0; i < n; ++i) nf += !u_known[i];
    double *b = (double *)calloc((size_t)nf + 1, sizeof(double));
    orc_csr *A = orc_reduce_system(K, u_known, u_in, f_in, b);
    double *xs = (double *)calloc((size_t)nf + 1, sizeof(double));
    double cost = 0.0;
    int64_t it;
    if (precond == 0) {
        it = orc_cg(A, b, stop_mode, tol, max_iter, xs, &cost, history, hist_len);
    } else {
        float *minv = (float *)malloc(sizeof(float) * 3 * (size_t)N + 4);
        orc_block_jacobi(K, N, u_known, precond, minv);
        double *ms = (double *)calloc((size_t)nf + 1, sizeof(double));
        double *mc = (double *)calloc((size_t)nf + 1, sizeof(double));
        int64_t *partner = (int64_t *)malloc(sizeof(int64_t) * ((size_t)nf + 1));
        int64_t f = 0;
        for (int64_t i = 0; i < N; ++i) {
            const int fx = !u_known[2 * i], fy = !u_known[2 * i + 1];
            if (fx) {
                ms[f] = (double)minv[3 * i];
                mc[f] = (double)minv[3 * i + 1];
                partner[f] = fy ? f + 1 : -1;
                ++f;
            }
            if (fy) {
                ms[f] = (double)minv[3 * i + 2];
                mc[f] = (double)minv[3 * i + 1];
                partner[f] = fx ? f - 1 : -1;
                ++f;
            }
        }
        it = orc_pcg(A, b, ms, mc, partner, stop_mode, tol, max_iter, xs, &cost, history, hist_len);
        free(minv);
        free(ms);
        free(mc);
        free(partner);
    }
    if (st) {
        st->iterations = it;
        st->final_cost = cost;
        st->n_free = nf;
        st->nnz_ff = A->nnz;
    }
    orc_csr_free(A);
    free(b);
    int64_t cur = 0;
    for (int64_t i = 0; i < n; ++i) u_out[i] = u_known[i] ? u_in[i] : xs[cur++];
    free(xs);
    for (int64_t i = 0; i < n; ++i) {
        if (!u_known[i]) {
            f_out[i] = f_in[i];
            continue;
        }
        double s = 0.0;
        for (int64_t p = K->rowptr[i]; p < K->rowptr[i + 1]; ++p) s += K->val[p] * u_out[K->col[p]];
        f_out[i] = s;
    }
    orc_csr_free(K);
    orc_stress(E, xy, conn, u_out, nu, youngs, stress_out);
    return 0;
}

int orc_run_sparse(int64_t N, int64_t E, const double *xy, const int32_t *conn,
                   const uint8_t *u_known, const double *u_in, const double *f_in, double youngs,
                   double nu, double thickness, int stop_mode, double tol, int64_t max_iter,
                   double *u_out, double *f_out, double *stress_out, orc_stats *st,
                   double *history, int64_t hist_len)
{
    return run_sparse_impl(N, E, xy, conn, u_known, u_in, f_in, youngs, nu, thickness, stop_mode, tol, max_iter,
                           u_out, f_out, stress_out, st, history, hist_len, 0);
}

/* precond: 1 Jacobi, 2 block-Jacobi (orc_block_jacobi); everything else as orc_run_sparse */
int orc_run_sparse_pcg(int64_t N, int64_t E, const double *xy, const int32_t *conn,
                       const uint8_t *u_known, const double *u_in, const double *f_in, double youngs,
                       double nu, double thickness, int stop_mode, double tol, int64_t max_iter,
                       double *u_out, double *f_out, double *stress_out, orc_stats *st,
                       double *history, int64_t hist_len, int precond)
{
    return run_sparse_impl(N, E, xy, conn, u_known, u_in, f_in, youngs, nu, thickness, stop_mode, tol, max_iter,
                           u_out, f_out, stress_out, st, history, hist_len, precond);
}
