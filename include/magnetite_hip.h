/*
 * magnetite_hip.h -- C ABI of the MI355X-native Magnetite solver hot path.
 *
 * This is the boundary a Rust `extern "C"` shim inside Magnetite's
 * `solver::run` binds (INTEGRATION.md shows the shim).  Every entry point
 * cites the reference interface it replaces (file:line into
 * kyle-tennison/Magnetite @ 2024_08_07).  Plain pointers and sizes only: no
 * C++ types, no exceptions, no torch types cross this boundary.
 *
 * Data model across the boundary (Rust `Vec<Node>` / `Option<f64>` are not
 * FFI-safe, so the shim flattens them -- datatypes.rs:1-29):
 *   xy[2N]       f64  node.vertex.{x,y}, interleaved              datatypes.rs:1-5
 *   conn[3E]     i32  element.nodes[0..3]                         datatypes.rs:17-18
 *   u_known[2N]  u8   1: node.ux/uy is Some (displacement prescribed, force unknown)
 *                     0: node.fx/fy is Some (force prescribed, displacement unknown)
 *   u_in[2N]     f64  prescribed displacement, read where u_known==1
 *   f_in[2N]     f64  prescribed force,        read where u_known==0
 *   DOF index    2*node + {0:x, 1:y}                              solver.rs:306-307,346-351
 * Every DOF has exactly one of (u,f) known -- the mesher guarantees it
 * (mesher.rs:615-624,881-900); the reference panics otherwise (solver.rs:431).
 * The shim checks it while flattening and reports MAG_ERR_BC_MISMATCH.
 *
 * Threading: one in-flight call per mag_ctx; calls block until the result is
 * complete; distinct contexts may be used from distinct threads.
 * Ownership: the caller owns every buffer it passes; the library keeps no
 * pointer after a call returns (device-resident inputs passed with
 * MAG_MEM_DEVICE are copied into context-owned buffers by mag_upload).
 * Errors: int status (0 == MAG_OK) + mag_last_error(); never aborts/throws.
 * The shim maps nonzero to MagnetiteError::Solver(msg) (error.rs:3-22).
 */
#ifndef MAGNETITE_HIP_H
#define MAGNETITE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAG_ABI_VERSION 4 /* 3: mag_stats gained exchange_timeout, best_param_mismatch; 4: edge_blocks (and, in the
                             reserved word behind it, tiles_per_workgroup) */

/* solver.rs:17-19 */
#define MAG_DOF 2
#define MAG_MAX_CG_ITER 10000000LL
#define MAG_TARGET_CG_COST 1e-4

typedef struct mag_ctx mag_ctx;

enum mag_status {
    MAG_OK = 0,
    MAG_ERR_BAD_ARGS = 1,
    MAG_ERR_BC_MISMATCH = 2,   /* no unknown displacement / inconsistent BC set (solver.rs:431 panics) */
    MAG_ERR_NOT_CONVERGED = 3, /* CG broke down: non-finite residual.  (The iteration cap is NOT an error: argmin's
                                  MaxItersReached is a normal termination and solver.rs:149-176 returns Ok(best_param);
                                  mag_run then returns MAG_OK with stats.converged = 0, stats.termination =
                                  MAG_TERM_MAX_ITERS and the lowest-cost iterate.  solver.rs:160-166 maps only an
                                  argmin Err, solver.rs:167-174 a missing best_param, to MagnetiteError::Solver.)   */
    MAG_ERR_HIP = 4,
    MAG_ERR_RCCL = 5,
    MAG_ERR_TOO_LARGE = 6,     /* an index would not fit int32                                   */
    MAG_ERR_STATE = 7          /* call order: no problem uploaded / not run yet                   */
};

/* CG stop rule.  argmin 0.10 reports cost = f(r.r) and the reference stops on the
 * ABSOLUTE threshold 1e-4 (solver.rs:153-154); whether f is sqrt or identity
 * cannot be verified offline, so both exist; default = the tighter. */
enum mag_stop {
    MAG_STOP_RNORM = 0,    /* sqrt(r.r) <= tol            (reference-compatible default) */
    MAG_STOP_RNORM_SQ = 1, /* r.r       <= tol                                            */
    MAG_STOP_REL = 2       /* sqrt(r.r) <= tol*sqrt(b.b)  (BASELINE config 3: "CG to 1e-8") */
};

/* Why the CG stopped (argmin's TerminationReason as the reference's Executor can produce it, solver.rs:149-157). */
enum mag_termination {
    MAG_TERM_NONE = 0,        /* no solve yet                                                              */
    MAG_TERM_TARGET_COST = 1, /* best_cost <= target_cost (solver.rs:154)                                  */
    MAG_TERM_MAX_ITERS = 2,   /* iter >= max_iters (solver.rs:153): the BEST iterate is returned            */
    MAG_TERM_BREAKDOWN = 3    /* non-finite residual (no reference counterpart: argmin would iterate on NaN) */
};

enum mag_operator {
    MAG_OP_MATRIX_FREE = 0, /* element-loop operator, K never read in the CG loop */
    MAG_OP_CSR = 1          /* reference-faithful: K_ff in CSR, solver.rs:31-36     */
};

enum mag_memory { MAG_MEM_HOST = 0, MAG_MEM_DEVICE = 1 };

typedef struct mag_options {
    int32_t device;       /* HIP device ordinal                                            */
    int32_t stop_mode;    /* enum mag_stop, default MAG_STOP_RNORM                         */
    double tol;           /* default MAG_TARGET_CG_COST (solver.rs:19)                     */
    int64_t max_iter;     /* default MAG_MAX_CG_ITER   (solver.rs:18)                      */
    int32_t cg_operator;  /* enum mag_operator, default MAG_OP_MATRIX_FREE                 */
    int32_t assemble_csr; /* 1 (default): build K in CSR as the reference does and take
                             the RHS and reactions from its rows; 0: matrix-free everywhere */
    int32_t check_every;  /* CG iterations per host convergence poll (default 64, even)    */
    int32_t use_graph;    /* 1 (default): replay the CG iteration block as a hipGraph      */
    int32_t tile_nodes;   /* owned nodes per workgroup tile: 256 | 512 | 1024; 0 (default): 512 for meshes of
                             >= 262144 nodes and for meshes of 32768..524288 nodes the on-chip CG can hold
                             (cg_variant 2), else 256                                       */
    int32_t history_len;  /* keep the cost of the first history_len iterations (tests)     */
    int32_t verbose;      /* 1: print the reference's "info:" phase lines to stdout        */
    int32_t op_variant;   /* 0 (default): LDS-halo operator when every tile fits LDS, else the
                             global-gather operator; 1: always the global-gather operator  */
    int32_t cg_variant;   /* One recurrence for 1 and 2 -- argmin's, with the numerator of beta, |r_new|^2, expanded
                             from exact dots of the previous iterate (r.r + 2 alpha r.q + alpha^2 q.q) so that one
                             grid-wide reduction per iteration suffices; alpha, the stop test and the reported
                             cost use the true r.r.
                             2 (default): on-chip -- when the whole mesh fits the registers and LDS of the chip
                               (about 0.5M nodes) the entire solve is ONE launch, the CG state never moves through
                               HBM, workgroups meet at a grid barrier once per iteration; otherwise as 1.
                             1: streaming -- one fused launch per CG iteration.
                             0: two launches per iteration, argmin's recurrences to the letter              */
    int32_t precision;    /* 0 (default): fp64 everywhere, as the reference.  1: the CG state, the operator and
                             (tile-relative) coordinates in fp32, dot products accumulated in fp64 -- the fp32
                             leg of BASELINE config 5's tolerance sweep; cannot meet the 1e-8 parity bar.  One GPU or
                             several (streaming protocol: one all-reduce per iteration, exchange buffer in doubles)  */
    int32_t preconditioner; /* 0 (default): plain CG, the reference's iteration (solver.rs:142).  OPT-IN ADDITION with
                             no reference counterpart: 1 Jacobi, 2 block-Jacobi on the 2x2 node-
                             diagonal blocks of K_ff (inverse blocks kept in fp32).  Same stop rule on the true
                             residual norm, same solution within the tolerance, fewer iterations on graded meshes.
                             Needs the fused LDS iteration (cg_variant 1, fp64, matrix-free operator).           */
    int32_t reserved;
} mag_options;

/* Borrowed view of the caller's flattened Vec<Node>/Vec<Element>/ModelMetadata
 * (solver.rs:543-547 arguments). */
typedef struct mag_problem {
    int64_t num_nodes;
    int64_t num_elements;
    const double *xy;       /* 2N */
    const int32_t *conn;    /* 3E */
    const uint8_t *u_known; /* 2N */
    const double *u_in;     /* 2N */
    const double *f_in;     /* 2N */
    double youngs_modulus;  /* ModelMetadata, datatypes.rs:22-29; solver.rs:559-561 */
    double poisson_ratio;
    double part_thickness;
    int32_t memory; /* enum mag_memory: where the five arrays live */
    int32_t reserved;
} mag_problem;

/* Caller-allocated outputs: every node.ux,uy,fx,fy and element.stress becomes
 * Some(..) (solver.rs:476-482,532-533).  NULL members are skipped. */
typedef struct mag_result {
    double *u_out;      /* 2N */
    double *f_out;      /* 2N */
    double *stress_out; /* E  */
    int32_t memory;     /* enum mag_memory */
    int32_t reserved;
} mag_result;

typedef struct mag_stats {
    int64_t iterations; /* "finished conjugate gradient approximation in {} iterations", solver.rs:101-104 */
    double final_cost;  /* cost of the returned iterate under stop_mode */
    double rhs_norm;    /* sqrt(b.b) */
    int32_t converged;
    int32_t breakdown;  /* non-finite r.r */
    int64_t n_free;     /* unknown displacements */
    int64_t nnz;        /* scalar nnz of K (0 if not assembled) */
    int64_t num_tiles;
    int64_t ell_entries; /* (node,incident element) slots incl. padding */
    int64_t halo_nodes;  /* sum over tiles of nodes staged from other tiles */
    int32_t max_tile_halo;
    int32_t lds_operator; /* 1: LDS-halo operator ran, 0: global-gather fallback */
    int32_t cg_kernel;    /* what ran the CG: 0 two launches per iteration, 1 one fused launch per iteration,
                             2 on-chip single launch, 3 CSR operator, 4 fp32 leg */
    int32_t exchange;     /* several ranks, how they traded per iteration: 0 one rank, 1 one all-reduce (RCCL or the test
                             transport), 2 on-chip kernels through the inboxes, 3 streaming kernels through the inboxes */
    /* per-phase device time, HIP events on the context's stream, milliseconds */
    double ms_order;     /* Hilbert ordering + incidence + tile tables (symbolic, matrix-free op) */
    double ms_csr_symbolic;
    double ms_element;   /* K_e build, solver.rs:548-567 */
    double ms_assemble;  /* numeric gather into CSR, solver.rs:290-331 */
    double ms_bc;        /* RHS / partition, solver.rs:365-404,427-432 */
    double ms_cg;        /* solver.rs:435-441 timed region minus the dense->CSR scan */
    double ms_post;      /* scatter-back, reactions, stress */
    double ms_total;
    int64_t best_iteration; /* the iteration whose iterate is returned: argmin's best_param (solver.rs:167-174).  Equal to
                               `iterations` unless the solve stopped at the iteration cap                            */
    int32_t termination;    /* enum mag_termination */
    int32_t persist_timeout; /* 1: the on-chip CG kernel gave up at its grid barrier in this run (not every workgroup was
                               co-resident) and the streaming kernels redid the solve; the context then streams for a
                               number of solves (8, doubling after every further failure, at most 1024) before it tries
                               the on-chip kernel again */
    int32_t exchange_timeout; /* 1: several ranks, the inbox exchange of the streaming kernels gave up in this run and the
                                 ranks redid the solve with one all-reduce per iteration; the context keeps the all-reduce
                                 until its inboxes are created anew (mag_comm_inbox_create)                        */
    int32_t best_param_mismatch; /* iteration cap only: the iterate of `best_iteration` is recovered by repeating the
                                    solve up to that iteration; 1 when the repeat took another kernel or exchange than the
                                    first pass, or its cost there is not the recorded best cost bit for bit --
                                    `final_cost` is then the cost of the iterate actually returned                  */
    int32_t edge_blocks;    /* cg_kernel 2 only: 1 when the on-chip kernel ran its edge-block instantiation (every node's
                               triangles folded into at most six symmetric 2 x 2 blocks held in registers: meshes whose
                               nodes all carry one fan of at most six triangles, closed, or five, open); 2 when it ran the
                               edge-block instantiation WITH OVERFLOW (rows that are one fan of any length -- gmsh-type
                               meshes: the blocks beyond six per node in an LDS pool); 0 when it walked the
                               triangles (nodes with several fans, or a pool that does not fit)                     */
    int32_t tiles_per_workgroup; /* cg_kernel 2 only: tiles each workgroup of the on-chip kernel held (1-4 of 512 nodes; up
                                    to three on one GPU run the instantiation with that many node slots per lane)      */
} mag_stats;

/* ---- lifecycle ------------------------------------------------------- */
int mag_version(void);
void mag_default_options(mag_options *opt);
/* NULL opt => defaults.  Returns NULL only if the context itself cannot be allocated. */
mag_ctx *mag_create(const mag_options *opt);
void mag_destroy(mag_ctx *ctx);
const char *mag_last_error(const mag_ctx *ctx);

/* ---- the drop-in entry: replaces solver::run, solver.rs:543-586 ------ */
/* upload -> run -> download in one blocking call. */
int mag_solve(mag_ctx *ctx, const mag_problem *problem, mag_result *result);

/* The same, split so a caller (bench.py) can keep inputs resident in HBM:
 *   mag_upload   copies the problem into context-owned device buffers
 *   mag_run      K_e build, assembly, BC elimination, CG, reactions, stress (solver.rs:548-583)
 *   mag_download copies u/f/stress out */
int mag_upload(mag_ctx *ctx, const mag_problem *problem);
int mag_run(mag_ctx *ctx);
int mag_download(mag_ctx *ctx, mag_result *result);
int mag_get_stats(const mag_ctx *ctx, mag_stats *stats);
/* cost of CG iterations 1..n (n <= options.history_len, <= iterations) */
int mag_get_history(mag_ctx *ctx, double *history, int64_t n);

/* ---- pieces of the path, exposed for parity tests -------------------- */
/* solver.rs:187-193 compute_element_area (pub; the mesher imports it, mesher.rs:9,523). Host-side. */
double mag_compute_element_area(const double *xy, const int32_t *tri);
/* solver.rs:204-230 compute_strain_displacement_matrix (pub): B[18], 3x6 row major, entries divided by 2*area. */
void mag_compute_strain_displacement_matrix(const double *xy, const int32_t *tri, double element_area, double *B);
/* solver.rs:240-250 compute_stress_strain_matrix (pub): D[9], 3x3 row major, plane stress. */
void mag_compute_stress_strain_matrix(double poisson_ratio, double youngs_modulus, double *D);
/* solver.rs:263-278 for every element of the uploaded problem: ke_out[36E] host, row-major 6x6. */
int mag_element_stiffness(mag_ctx *ctx, double *ke_out);
/* solver.rs:290-331: K (2N x 2N) in CSR, ascending columns, structural pattern.
 * Call once with all outputs NULL to get nnz, then with host buffers rowptr[2N+1], col[nnz], val[nnz]. */
int mag_assemble_csr(mag_ctx *ctx, int64_t *nnz, int32_t *rowptr, int32_t *col, double *val);
/* solver.rs:365-404,427-432,123-137: K_ff with exact zeros dropped + b, compact unknown numbering.
 * Same two-call pattern: rowptr[n_free+1], col[nnz_ff], val[nnz_ff], b[n_free]. */
int mag_reduce_system(mag_ctx *ctx, int64_t *n_free, int64_t *nnz_ff, int32_t *rowptr, int32_t *col,
                      double *val, double *b);
/* y = K x with the matrix-free element-loop operator on the uploaded mesh
 * (x, y: host, 2N, caller's DOF numbering).  masked != 0 applies M K M with M
 * zeroing prescribed-displacement DOFs (that is K_ff embedded in full length). */
int mag_apply_operator(mag_ctx *ctx, const double *x, double *y, int32_t masked);
/* Bench helper: `reps` back-to-back launches of the CG iteration kernel (cg_variant 1: the fused
 * iteration kernel; 0: the operator kernel of the two-launch iteration) on the context's stream
 * between two HIP events; *ms_per_launch = elapsed / reps.  Needs an uploaded problem, not a solve: straight after
 * mag_upload the symbolic phase is run and the vectors are zeroed.  With several ranks the launch covers the tiles
 * this rank owns (its per-GPU share). */
int mag_time_operator(mag_ctx *ctx, int32_t reps, double *ms_per_launch);
/* The same for the plain matrix-free SpMV y = M K M v (no CG update fused in). */
int mag_time_spmv(mag_ctx *ctx, int32_t reps, double *ms_per_launch);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ------------------- */
#define MAG_UNIQUE_ID_BYTES 128
/* rank 0 calls this and broadcasts the bytes out of band (bench.py: torch.distributed) */
int mag_comm_get_unique_id(void *id_out);
int mag_comm_init_rccl(mag_ctx *ctx, const void *unique_id, int32_t nranks, int32_t rank);
/* What the context's communicator is: info[0] = ranks, info[1] = this rank, info[2] = transport (0 none, 1 RCCL,
 * 2 host callback), info[3] = ranks as RCCL itself reports them (ncclCommCount; 0 without an RCCL communicator). */
int mag_comm_query(const mag_ctx *ctx, int32_t info[4]);
/* test transport: sum-all-reduce of a host buffer supplied by the caller (gloo in tests/) */
typedef int (*mag_allreduce_fn)(void *user, double *host_buf, int64_t count);
int mag_comm_init_callback(mag_ctx *ctx, int32_t nranks, int32_t rank, mag_allreduce_fn fn, void *user);
/* EXPERIMENTAL multi-GPU on-chip CG (correct, but slower than the default once a rank has more than a few hundred
 * interface nodes: every granule is its own PCIe transaction -- see DESIGN.md): a window of HOST memory that every
 * rank of the node has mapped at `host_ptr` (the same
 * physical pages: POSIX shared memory, bytes >= 64 + 128 * nranks + 64 * interface nodes; a few MB is plenty).  With a
 * window and cg_variant 2 each rank runs its share of the mesh as ONE persistent launch and the per-iteration
 * exchange (one record of sums per rank, q of the interface nodes) goes through the window as tagged granules instead
 * of a collective per iteration; the communicator set by mag_comm_init_* is still used to line the launches up, to
 * agree on a fallback and to assemble the solution.  Without a window, or when the mesh does not fit the chips, the
 * streaming kernels + one all-reduce per iteration run.  NULL / 0 removes the window.  The window needs no preparation:
 * the extent a solve uses is cleared by rank 0 before the ranks line up, every solve. */
int mag_comm_set_window(mag_ctx *ctx, void *host_ptr, uint64_t bytes);
/* The same protocol with the window where it belongs: one INBOX per rank in that rank's device memory, mapped by the
 * other ranks through HIP IPC.  A rank only reads its own inbox (polls stay in local HBM); writers store into the
 * inboxes of the ranks that read a value (across xGMI).  Every rank: mag_comm_inbox_create(ctx, bytes, handle) (bytes
 * as for the window; `handle` receives MAG_IPC_HANDLE_BYTES bytes), exchange the handles by any means, then
 * mag_comm_inbox_open(ctx, all_handles) with the nranks handles in rank order.  bytes = 0 removes the inboxes.
 * When the mesh does not fit the chips the streaming kernels keep running one launch per iteration and trade through the
 * same inboxes between launches (mag_stats.exchange = 3) instead of one all-reduce per iteration.
 * Each rank's on-chip launch ends with one exchange workgroup (no tile: it gathers the rank's partial sums, trades them with the
 * other ranks through the inboxes and republishes the total) whenever a CU is free for it.
 * Exercised with up to eight ranks on ONE GPU only (~10 us per CG iteration with 8 ranks against ~40-90 through host
 * memory): measure before relying on it on a node, as bench.py does. */
#define MAG_IPC_HANDLE_BYTES 64
int mag_comm_inbox_create(mag_ctx *ctx, uint64_t bytes, void *handle_out);
int mag_comm_inbox_open(mag_ctx *ctx, const void *handles);

#ifdef __cplusplus
}
#endif
#endif /* MAGNETITE_HIP_H */
