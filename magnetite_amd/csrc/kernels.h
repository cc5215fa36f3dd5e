// Internal: launch wrappers of the hand-written gfx950 kernels.
//   symbolic.hip  integer kernels of the one-time symbolic phases
//   exact.hip     reference-order fp64 arithmetic (compiled -ffp-contract=off)
//   cg.hip        matrix-free element-loop operator + fused CG kernels
#pragma once
#include <cstdint>

#include <hip/hip_runtime.h>

// Wait states of the hand-written VMEM stores (persist.hip, exact.hip).  The compiler's hazard recognizer does not look at the
// uses inside an inline-asm statement, nor at what the asm's last instruction needs from the code behind it, so the asm
// strings carry them: five wait states before a store addressed through an SGPR base (a VALU write of that SGPR -- a spill
// coming back through v_readlane_b32 -- may sit right in front of the asm), two after a store of more than 8 bytes (its data
// registers must not be rewritten before).  scripts/isa_lint.py checks both on the emitted ISA (tests/test_isa_hazards.py);
// -DMAG_ASM_NO_WAITSTATES drops them, for that test only.
#ifdef MAG_ASM_NO_WAITSTATES
#define MAG_WS_SBASE ""
#define MAG_WS_DATA ""
#else
#define MAG_WS_SBASE "s_nop 4\n\t"
#define MAG_WS_DATA "\n\ts_nop 1"
#endif

namespace magk {

constexpr int kMaxGrid = 1024;  // capacity of the dot-partial arrays == upper bound on workgroups of a CG kernel
constexpr int kHilbertBits = 16;
constexpr int kMaxLdsNodes = 2016; // (owned + halo) nodes per tile the LDS-halo operator stages: 32 B each < 64 KiB

// ------------------------------------------------------------ symbolic.hip
// bbox[4] = {xmin, ymin, xmax, ymax}; scratch >= 4*256 doubles
void bbox(const double *xy, int64_t N, double *scratch, double *bbox4, hipStream_t s);
// keys[i] = Hilbert index of node i on a 2^16 x 2^16 lattice over the bbox (uniform scale), ids[i] = i
void hilbert_keys(const double *xy, int64_t N, const double *bbox4, uint32_t *keys, uint32_t *ids, hipStream_t s);
// perm = new->old (sorted ids).  Writes iperm (old->new), xyP[new] and maskP[new] = known_x | known_y<<1
void apply_order(const uint32_t *perm, const double *xy, const uint8_t *u_known, int64_t N, int32_t *iperm,
                 double *xyP, uint8_t *maskP, int32_t *known_count, const int32_t *cdeg, int32_t *deg, const double *f_in,
                 double *bP, hipStream_t s);
// per element corner k=3e+c: keys[k] = iperm[conn[k]], vals[k] = k, deg[key]++ ; out-of-range conn sets *err
void incidence_keys(const int32_t *conn, int64_t E, const int32_t *iperm, int64_t N, uint32_t *keys,
                    uint32_t *vals, int32_t *deg, int32_t *err, hipStream_t s);
// several ranks, sharded ordering phase (symbolic.hip): tiles whose tables this rank needs; incidence pairs of their nodes only;
// the interface list and its reader masks from one pass over the elements
struct RankTiles {
    int32_t lo[9]; // first tile of rank r (lo[R] = T)
    int32_t R;
};
void need_tiles(const int32_t *conn, int64_t E, const int32_t *iperm, const uint8_t *maskP, int64_t N, int32_t B, int32_t T,
                int32_t t0, int32_t t1, bool prescribed_rows, uint8_t *need, const RankTiles &rt, uint8_t *readers, hipStream_t s);
void incidence_flags(const int32_t *conn, int64_t E, const int32_t *iperm, int64_t N, int32_t B, const uint8_t *need,
                     int32_t *flag, int32_t *err, hipStream_t s);
void incidence_emit(const int32_t *conn, int64_t E, const int32_t *iperm, int64_t N, const int32_t *off, uint32_t *keys,
                    uint32_t *vals, int32_t *deg, hipStream_t s);
void zero_unneeded_deg(const uint8_t *need, int64_t N, int32_t B, int32_t *deg, hipStream_t s);
void iface_flags(const uint8_t *readers, int64_t N, int32_t *flag, hipStream_t s);
void iface_emit(const uint8_t *readers, const int32_t *off, int64_t N, int32_t *iface, uint8_t *iface_readers, hipStream_t s);
// tile_deg[t] = max deg over the tile's nodes; tile_cnt[t] = tile_deg[t] * B  (int64)
void tile_degree(const int32_t *deg, int64_t N, int32_t B, int32_t T, int32_t *tile_deg, int64_t *tile_cnt,
                 hipStream_t s);
// ELL fill: slot k of node i (tile t, lane l) at ell[tile_off[t] + k*B + l] = (b, c) Hilbert ids of the
// incident element's other two corners in cyclic order after i; (-1,-1) pads.
void fill_ell(const int32_t *inc_off, const uint32_t *inc, const int32_t *conn, const int32_t *iperm,
              const int32_t *tile_deg, const int64_t *tile_off, int64_t N, int32_t B, int32_t T, int2 *ell,
              hipStream_t s);
// tile-local numbering: per node count of references to nodes of other tiles (cnt[N] = 0 sentinel) ...
void halo_count(const int32_t *inc_off, const uint32_t *inc, const int32_t *conn, const int32_t *iperm, int64_t N,
                int32_t B, int32_t *cnt, hipStream_t s);
// ... their keys (tile<<32 | node) at off[i]..., to be sorted and uniqued ...
void halo_emit(const int32_t *inc_off, const uint32_t *inc, const int32_t *conn, const int32_t *iperm, int64_t N,
               int32_t B, const int32_t *off, uint64_t *keys, hipStream_t s);
// ... halo_g[blk] = node of each unique key, tile_hcnt[tile]++ ...
void halo_unique(const uint64_t *keys, const int32_t *head, const int32_t *blk, int64_t n, int32_t *halo_g,
                 int32_t *tile_hcnt, hipStream_t s);
// ... and the ELL table with 16-bit tile-local ids: word = lb | lc<<16, 0xffffffff pads; owned l < B,
// halo B + rank in halo_g[tile_hoff[t]..tile_hoff[t+1])
// within-tile order by valence (symbolic.hip): triangles per node in caller numbering, then each tile of the Hilbert order
// stably partitioned by max(0, triangles - 6) -- the identity on tiles whose nodes all have at most six triangles
void count_degree(const int32_t *conn, int64_t E, int64_t N, int32_t *cdeg, hipStream_t s);
void tile_valence_partition(const uint32_t *perm_in, const int32_t *cdeg, int64_t N, int32_t B, int32_t T, uint32_t *perm_out,
                            hipStream_t s);
void fill_ell16(const int32_t *inc_off, const uint32_t *inc, const int32_t *conn, const int32_t *iperm,
                const int32_t *tile_deg, const int64_t *tile_off, const int32_t *tile_hoff, const int32_t *halo_g,
                int64_t N, int32_t B, int32_t T, uint32_t *ell, uint32_t *ell_asm, uint16_t *ell_pos, hipStream_t s);
// halo_xy[i] = xyP[halo_g[i]]
void halo_coords(const int32_t *halo_g, const double *xyP, int64_t n, double *halo_xy, hipStream_t s);
// CSR pattern: 9 (row node, col node) pairs per element, key = row<<32 | col, val = 9e + 3a + b
void csr_pairs(const int32_t *conn, int64_t E, uint64_t *keys, uint32_t *vals, hipStream_t s);
// CSR pattern straight from the incidence lists: rowcnt[i] = distinct nodes of node i's incident elements (0 where
// local is given and local[i] == 0; rowcnt[N] = 0), *overflow = 1 if a row exceeds the register array; then, with
// bptr = scan(rowcnt), bcol[bptr[i] ..] = those nodes, ascending
void pattern_count(const int32_t *inc_off, const uint32_t *inc, const uint32_t *perm, const int32_t *conn,
                   const uint8_t *local, int64_t N, int32_t *rowcnt, int32_t *overflow, const uint8_t *need, int32_t B,
                   hipStream_t s);
void pattern_fill(const int32_t *inc_off, const uint32_t *inc, const uint32_t *perm, const int32_t *conn,
                  const uint8_t *local, int64_t N, const int32_t *bptr, int32_t *bcol, const uint8_t *u_known,
                  uint8_t *touch, const uint8_t *need, int32_t B, hipStream_t s);
// multi-GPU: local[i] = 1 for the nodes whose K rows this rank keeps (owned, one ghost layer, prescribed nodes) ...
void mark_local(const uint32_t *perm, const uint8_t *maskP, int64_t N, int32_t own0, int32_t own1,
                const int32_t *halo_g, int32_t h0, int32_t h1, uint8_t *local, hipStream_t s);
// ... cnt[e] = pairs element e contributes to local rows (cnt[E] = 0), and the pairs themselves, compacted at off[e]
void csr_pair_count(const int32_t *conn, int64_t E, const uint8_t *local, int32_t *cnt, hipStream_t s);
void csr_pairs_local(const int32_t *conn, int64_t E, const uint8_t *local, const int32_t *off, uint64_t *keys,
                     uint32_t *vals, hipStream_t s);
// head[k] = 1 where sorted key k starts a new (row,col) block
void csr_heads(const uint64_t *keys, int64_t n, int32_t *head, hipStream_t s);
// for heads: seg_start[blk] = k, brow/bcol[blk] = row/col node, rowcnt[row node]++   (blk = exclusive scan of head)
void csr_segments(const uint64_t *keys, const int32_t *head, const int32_t *blk, int64_t n, int32_t *seg_start,
                  int32_t *brow, int32_t *bcol, int32_t *rowcnt, hipStream_t s);
// scalar CSR arrays of K from the node-block pattern
void csr_export(const int32_t *bptr, const int32_t *bcol, int64_t N, int32_t *rowptr, int32_t *col, hipStream_t s);
// free-DOF numbering flags: isfree[i] = !u_known[i]
void free_flags(const uint8_t *u_known, int64_t n, int32_t *isfree, hipStream_t s);
// K_ff row counts (exact zeros dropped) and fill, compact numbering
void reduce_count(const int32_t *bptr, const int32_t *bcol, const double *kval, const uint8_t *u_known, int64_t N,
                  int32_t *cnt /*2N*/, hipStream_t s);
void reduce_fill(const int32_t *bptr, const int32_t *bcol, const double *kval, const uint8_t *u_known,
                 const int32_t *fidx, const int32_t *rowoff /*2N, by full row*/, int64_t N, int32_t *rowptr_ff,
                 int32_t *col_ff, double *val_ff, hipStream_t s);

// --------------------------------------------------------------- exact.hip
// solver.rs:263-278 per element, ke[36e + 6i + j]
void element_stiffness(const double *xy, const int32_t *conn, int64_t E, double nu, double youngs, double thick,
                       double *ke, hipStream_t s);
// solver.rs:290-331 as an atomic-free gather: one thread per (row node, col node) block sums its segment of
// sorted pairs in ascending element order into the four scalar CSR slots.
void assemble_gather(const uint64_t *keys, const uint32_t *vals, const int32_t *seg_start, int64_t nb,
                     const int32_t *bptr, const double *ke, double *kval, hipStream_t s);
// the same assembly without the K_e buffer: per (row,col) block, only the needed 2x2 blocks of the incident elements
void assemble_rows(const int32_t *brow, const int32_t *bcol, const int32_t *bptr, int64_t nb, const int32_t *inc_off,
                   const uint32_t *inc, const int32_t *iperm, const int32_t *conn, const double *xy, double nu,
                   double youngs, double thick, double *kval, hipStream_t s);
// the same per element tile (kAsmNodes nodes of the Hilbert order + their incident elements, one workgroup each): element
// areas and the divided B entries staged once per (node, element) entry in LDS, blocks gathered from LDS (default)
void assemble_tiles(const int32_t *bcol, const int32_t *bptr, const int32_t *inc_off, const uint32_t *inc,
                    const uint32_t *perm, const int32_t *conn, const double *xy, int64_t N, double nu, double youngs,
                    double thick, double *kval, hipStream_t s);
// the same assembly fed from the CG tiles (exact.hip, k_assemble_fan): coordinates and caller ids staged in LDS, the
// per-node corner words of fill_ell16 (ell_asm) instead of the incidence -> connectivity -> coordinate gathers; false
// when a tile's image does not fit one CU's LDS (the caller then uses assemble_tiles)
size_t assemble_ctiles_lds(int32_t B, int32_t cap); // dynamic LDS of k_assemble_fan for B-node tiles with an image of `cap` nodes
bool assemble_ctiles(const int32_t *bcol, const int32_t *bptr, const uint32_t *perm, const double *xyP,
                     const double *halo_xy, const int32_t *tile_hoff, const int32_t *tile_deg, const int64_t *tile_off,
                     const uint32_t *ell_asm, const uint16_t *ell_pos, const int32_t *inc_off, const uint32_t *inc,
                     const int32_t *conn, const double *xy, int64_t N, int32_t B, int32_t T, int32_t cap, double nu,
                     double youngs, double thick, double *kval, hipStream_t s);
// apply_order has written b = 0.0 + f (0 on prescribed DOFs) for every node; this redoes the rows with a prescribed column
// (touch: the pattern kernel's flags, Hilbert order)
void rhs_touched(const int32_t *bptr, const int32_t *bcol, const double *kval, const uint8_t *u_known, const double *u_in,
                 const double *f_in, const uint32_t *perm, const uint8_t *touch, int64_t N, double *bP, hipStream_t s);
// solver.rs:365-404,427-432: b[row] = sum_{known cols, ascending} -(K*u) + f  (0 on prescribed rows),
// written in Hilbert order: bP[2*iperm[node]+a]
void rhs_from_csr(const int32_t *bptr, const int32_t *bcol, const double *kval, const uint8_t *u_known,
                  const double *u_in, const double *f_in, const uint32_t *perm, uint8_t *touch, bool touch_ready,
                  int64_t N, double *bP, hipStream_t s);
// same, compact numbering (for mag_reduce_system)
void rhs_compact(const int32_t *bptr, const int32_t *bcol, const double *kval, const uint8_t *u_known,
                 const double *u_in, const double *f_in, const int32_t *fidx, int64_t N, double *b, hipStream_t s);
// solver.rs:443-454: u[2*perm[i]+a] = known ? u_in : xP[2i+a]
void scatter_back(const double *xP, const uint32_t *perm, const uint8_t *u_known, const double *u_in, int64_t N,
                  double *u, hipStream_t s);
// solver.rs:456-469: f[i] = known ? K_row(i).u : f_in[i]
void reactions_from_csr(const int32_t *bptr, const int32_t *bcol, const double *kval, const uint8_t *u_known,
                        const double *u, const double *f_in, int64_t N, double *f, hipStream_t s);
// matrix-free variant: f = known ? yP[iperm] : f_in
void reactions_from_apply(const double *yP, const int32_t *iperm, const uint8_t *u_known, const double *f_in,
                          int64_t N, double *f, hipStream_t s);
// solver.rs:496-535
void element_stress(const double *xy, const int32_t *conn, const double *u, int64_t E, double nu, double youngs,
                    double *stress, hipStream_t s);

// ------------------------------------------------------------------ cg.hip
struct CgState {
    double rr_hist[2]; // r.r of iterations k (slot k&1) and k-1
    double target;     // absolute threshold on the cost
    double final_cost;
    double bb;         // b.b
    double alpha_last;
    long long iterA;   // written by the update kernel, read by the operator kernel
    long long iterB;   // written by the operator kernel, read by the update kernel
    long long iterations;
    long long max_iter;
    int done;
    int converged;
    int breakdown;
    int stop_mode;
    double tol;
    double best_cost;    // lowest cost over iterations 1..k and the iteration that had it: argmin's best_param
    long long best_iter; // (solver.rs:167-174 returns state.best_param, not the last iterate)
    double pad[1];
};

struct OpParams {
    int64_t N;
    int32_t T; // tiles of the whole mesh
    int32_t nPart;
    int32_t t0, t1;   // tile range this rank owns (0, T on one GPU)
    int32_t own0, own1; // node range of those tiles
    const int32_t *iface; // multi-GPU: sorted Hilbert ids of every node some rank reads but does not own
    int32_t n_iface;
    int32_t pad1;
    const double2 *xyP;
    const uint8_t *maskP;
    const int32_t *tile_deg;
    const int64_t *tile_off;
    const int2 *ell;         // gather variant: global (Hilbert) ids, halo read from global memory
    const uint32_t *ell16;   // LDS-halo variant (nullptr => gather variant): ring table of tile-local ids (k_ring16)
    const int32_t *tile_hoff; // T+1 offsets into halo_g
    const int32_t *halo_g;   // per tile: sorted Hilbert ids of its halo nodes
    const double2 *halo_xy;  // their coordinates, same layout (static copy)
    int32_t cap;             // LDS image capacity in nodes: B + max halo, <= kMaxLdsNodes
    int32_t wt;              // write-through (sc1) stores of pnew and q
    double c0, nu, h; // E t / (2 (1-nu^2)), nu, (1-nu)/2
    // CG mode
    const double2 *r;
    const double2 *pprev;
    double2 *pnew;
    double2 *q;
    double2 *x; // x += alpha_{k-1} p_{k-1} happens here, where p_{k-1} is read anyway
    const double *partRR;
    double *partPQ;
    CgState *st;
    double *hist;
    int32_t hist_len;
    // plain mode: y = [M] K [M] v
    int32_t masked;
    const double2 *v;
    double2 *y;
};

struct UpdParams {
    int64_t N;
    int32_t T;
    int32_t nPart;
    int32_t t0, t1;
    double2 *r;
    const double2 *q;
    const double *partPQ;
    double *partRR;
    CgState *st;
    int32_t wt;
};

// ---- reference-faithful CG on K_ff in CSR (MAG_OP_CSR): compact unknown numbering, solver.rs:31-36 ----
struct CsrCgParams {
    int64_t n;      // unknowns
    int32_t nPart;  // partials written by the previous launch
    int32_t hist_len;
    const int32_t *rowptr;
    const int32_t *col;
    const double *val;
    double *x;
    const double *r;
    const double *pprev;
    double *pnew;
    double *q;
    const double *partRR;
    double *partPQ;
    CgState *st;
    double *hist;
};
// p = -r + beta p_prev, x += alpha_prev p_prev (state machine of the two-launch variant)
void csr_p_launch(const CsrCgParams &P, hipStream_t s);
// q = K_ff p, row sums in ascending column order; p.q partials
void csr_spmv_launch(const CsrCgParams &P, hipStream_t s);
// r += alpha q; r.r partials (plain double arrays)
void csr_update_launch(int64_t n, double *r, const double *q, const double *partPQ, int32_t nPart, double *partRR,
                       CgState *st, hipStream_t s);
void csr_init(const double *b, double *r, int64_t n, double *partRR, hipStream_t s);
int csr_grid(int64_t n);
void expand_free(const double *xf, const int32_t *fidx, const uint8_t *u_known, const double *u_in, int64_t n2,
                 double *u, hipStream_t s);

// ---- fused single-launch CG iteration (cg_variant 1) ----
// One record per node keeps what a neighbouring tile must read of it in one place: r, q = A p, p.
struct Rqp {
    double2 r, q, p;
};
struct TileMeta { // one 32-byte scalar load per tile instead of four dependent ones
    int64_t ell_off;
    int32_t deg, hoff, nh, ent; // deg: ring words of the longest row, ent: its entries (2 deg - 1 or 2 deg)
    int64_t pad2;
};
struct FusedState {
    long long jslot[2]; // launch counter, read from slot[par], written to slot[par^1]
    double target, final_cost, bb, tol;
    long long iterations, max_iter;
    int done, converged, breakdown, stop_mode;
    double best_cost;    // as in CgState
    long long best_iter;
    int exchange_timeout; // multi-GPU, inbox exchange of the streaming kernels: a wait ran out (k_stream_exchange)
    unsigned int exchange_tag_base; // ... and the solve's sequence number << 24: tag of exchange e = base + e (host-written)
    double pad[3];
};
struct FusedParams {
    int64_t N;
    int32_t T, nPart, t0, t1, own0, own1, n_iface, cap, wt, par, hist_len, pad;
    const double2 *xyP;
    const uint8_t *maskP;
    const TileMeta *meta;
    const uint32_t *ell16;
    const int32_t *halo_g;
    const double2 *halo_xy;
    const int32_t *iface;
    double c0, nu, h;
    const Rqp *in;
    Rqp *out;
    double2 *x;
    const double *part_in; // 4 * nPart: sums of r.r, p.q, r.q, q.q of the previous iterate
    double *part_out;      // 4 * gridDim
    int32_t part_stride;    // entries per partial array in part_out
    int32_t part_stride_in; // ... and in part_in (1 when the sums arrive all-reduced)
    FusedState *st;
    double *hist;
    // multi-GPU exchange (all null on one GPU): see k_cg_fused
    const int32_t *own_qslot;  // N
    const int32_t *halo_qslot; // halo_total
    const double2 *comm_in_q;  // n_iface, all-reduced q of iterate j-1
    double2 *comm_out_q;       // n_iface, this rank's q of iterate j (zeros where it is not the owner)
    // opt-in preconditioner (null = plain CG, the reference's iteration): inverse 2x2 node blocks (i00, i01, i11, 0)
    // in fp32, Hilbert order, and their per-tile halo copies; the partial arrays are then five, not four
    const float4 *minvP;
    const float4 *halo_minv;
};
// rewrites the tile-local table of fill_ell16 into ring form in place (symbolic.hip, k_ring16)
// row_info (T * B bytes, may be null): per row the number of ring entries (bits 0-5), closed fan (bit 6), one fan (bit 7)
void ring16(const int32_t *tile_deg, const int64_t *tile_off, int32_t B, int32_t T, uint32_t *ell, int32_t *tile_rdeg, int32_t block_entries,
            uint8_t *row_info, hipStream_t s);
// edge blocks beyond the nb kept in registers: per-node counts (npad + 1 entries, the last 0) for the scan, then the limits
// the host checks -- out[0] the most records a workgroup of k tiles holds, out[1] the most of one node
void ovf_counts(const uint8_t *row_info, int64_t npad, int32_t nb, int32_t *cnt, hipStream_t s);
void ovf_limits(const int32_t *off, int32_t B, int32_t k, int32_t t0, int32_t t1, int32_t *out, hipStream_t s);
void tile_meta(const int32_t *tile_deg, const int32_t *tile_ent, const int64_t *tile_off, const int32_t *tile_hoff,
               int32_t T, TileMeta *meta, hipStream_t s);
// workgroups of the fused kernel: all co-resident (occupancy query x CUs), so the launch is one persistent round --
// measured best on MI355X (fewer, fatter workgroups also mean fewer dot partials for every workgroup to reduce)
int fused_grid(int32_t B, int32_t cap, int32_t tiles, bool comm, bool pre);
void fused_launch(const FusedParams &P, int32_t B, int32_t grid, hipStream_t s);
// in[node] = {-b, 0, 0}; part[0..] = {b.b partials over owned tiles, 1/grid, 0, 0} so that launch 0 gets
// alpha finite, beta = 1
void fused_init(const double2 *bP, const float4 *minvP, Rqp *in, Rqp *out, int64_t N, int32_t B, int32_t T, int32_t t0,
                int32_t t1, double *part, int32_t stride, int32_t grid, hipStream_t s);
void halo_minv(const int32_t *halo_g, const float4 *minvP, int64_t halo_total, float4 *out, hipStream_t s);
// exact.hip: inverse node-diagonal blocks of K_ff (kind 1: diagonal only, 2: 2x2 blocks), fp32, Hilbert order
void precond_blocks(const int32_t *inc_off, const uint32_t *inc, const uint32_t *perm, const int32_t *conn,
                    const double *xy, const uint8_t *u_known, int64_t N, double nu, double youngs, double thick, int kind,
                    float4 *minvP, hipStream_t s);
void fused_setup(const double *part, int32_t nPart, int32_t stride, int stop_mode, double tol, long long max_iter,
                 FusedState *st, hipStream_t s);
// multi-GPU: slot tables of the exchange buffer [4 x g_all dot partials | q of the n_iface interface nodes]
void comm_slots(const int32_t *iface, int32_t n_iface, int32_t own0, int32_t own1, const int32_t *halo_g,
                int64_t halo_total, int64_t N, int32_t *own_qslot, int32_t *halo_qslot, hipStream_t s);

// ---- on-chip (persistent) CG: the whole solve in one launch when every tile fits registers + LDS (persist.hip) ----
struct PersistParams {
    int64_t N;
    int32_t T, tiles_per_wg, cap, maxh, hist_len, stop_mode;
    uint32_t spin_limit; // polls of the arrival words before a workgroup gives up (sets the timeout word, leaves)
    uint32_t pad;
    long long max_iter;
    double tol, c0, nu, h;
    const double2 *xyP;
    const uint8_t *maskP; // bit 0/1: prescribed ux/uy, bit 2: read by some tile's halo (owner publishes q)
    const TileMeta *meta;
    const uint32_t *ell16;
    const int32_t *halo_g;
    const double2 *halo_xy;
    const double2 *bP;
    double2 *x;
    unsigned long long *qg;   // 2 * N * 4 granules {epoch, 32-bit half}: published q, by parity
    unsigned long long *recg; // 2 * grid * 8 granules: dot partials of every workgroup, by parity
    uint32_t *sync;           // [9] timeout word; zeroed, like the granules, before every launch
    FusedState *st;
    double *hist;
    // multi-GPU (MG instantiation; nranks == 1 otherwise): this rank runs tiles [t0, t1) and exchanges through a
    // window of host memory every rank has mapped (granules again, system scope): one record of sums per rank, q of the
    // interface nodes.  Tags carry a per-solve sequence number in their upper bits, so the window is never zeroed.
    int32_t t0, t1, rank, nranks, n_iface;
    uint32_t tag_base;          // solve sequence << 24
    const int32_t *own_qslot;   // N: interface slot of an owned node other ranks read, -1 otherwise
    const int32_t *halo_qslot;  // halo_total: interface slot of a halo entry another rank owns, -1 otherwise
    // The window, seen as one INBOX per rank (layout of each: 64 bytes {timeout word}, 2 * nranks * 8 record granules,
    // 2 * n_iface * 4 q granules).  A rank only ever reads its own inbox; writers store into the inbox of every rank
    // that reads the value.  Host-memory window: all inbox pointers are the same shared pages (win_shared = 1, one
    // store serves everybody).  Peer window: inbox[r] is rank r's device memory, IPC-mapped (stores cross xGMI, polls
    // stay in local HBM).
    uint8_t *inbox[8];
    int32_t win_shared;
    int32_t comm_wg; // 1: the grid ends with an exchange workgroup (persist_comm_loop); device inboxes only
    const uint8_t *iface_readers; // n_iface: bit r set when rank r reads the interface node of that slot
    unsigned long long *grec;    // device: 2 * 8 granules, the grid-wide sums republished by workgroup 0
    unsigned long long *stamps;  // diagnostic build (-DMAG_PERSIST_STAMPS) only: per workgroup, phase times in 10 ns ticks
    // edge-block instantiation: the nodes' symmetric 2 x 2 blocks, value c of node i at kblocks[c * kb_stride + i]
    // (persist.hip, k_edge_blocks: once per solve, before the launch)
    const double *kblocks;
    int64_t kb_stride;
    // ... with overflow (EBM == 2, meshes whose rows are single fans of any length -- gmsh-type meshes): blocks beyond the
    // six in registers live in LDS, 32-byte records {k11, k12, k22, ring entry}; k_edge_blocks writes them to ovf_rec at
    // ovf_off[node] + j (exclusive scan of the per-node counts over the padded Hilbert order), the on-chip kernel copies
    // its workgroup's run into its pool.  row_info: k_ring16's per-row byte (entries, closed, one fan).
    const uint8_t *row_info;
    const int32_t *ovf_off;
    double *ovf_rec;
    int32_t pool_cap; // records the LDS pool of a workgroup holds (the host checked every workgroup's run against it)
    int32_t pad3;
};
// multi-GPU, streaming kernels: the per-iteration exchange [dot partials | interface q] through the ranks' device
// inboxes instead of an all-reduce, in place on `buf` (persist.hip, k_stream_exchange)
// `fpar` = parity of the iteration launch that has just filled `buf`; the exchange's epoch (hence its inbox parity and
// its tag) is read from the launch counter that launch left in FusedState: no per-launch host value, so a block of
// [iteration launch, exchange] pairs replays from a hipGraph
void stream_exchange_launch(double *buf, int32_t g_all, int32_t n_iface, int32_t rank, int32_t nranks, int32_t own0,
                            int32_t own1, int32_t fpar, uint32_t spin_limit, const int32_t *iface,
                            const uint8_t *iface_readers, void *const *inboxes, FusedState *st, hipStream_t s);
int persist_threads(); // workgroup shape of the on-chip kernel: 512 (x 4 nodes per lane) or 768 (x 3); MAG_TUNE_PERSIST_THREADS
int persist_tiles_per_wg(int32_t B, int threads); // tiles one workgroup keeps on chip (0: tile size not supported)
// eb_mode: 0 triangle walk, 1 edge blocks (every row a fan of <= 6 blocks), 2 edge blocks with `pool` overflow records in LDS
size_t persist_lds_bytes(int32_t B, int32_t cap, int32_t maxh, int threads, int eb_mode = 0, int32_t pool = 0, bool mg = false);
// MG kernel when nranks > 1; eb_mode as above (the host decides from ring16's flags and the overflow limits)
void persist_launch(const PersistParams &P, int32_t B, int32_t grid, int threads, int eb_mode, hipStream_t s);
int persist_block_entries(); // block entries per node of that instantiation
// ... and its blocks: 3 * persist_block_entries() doubles per node of the T * B padded nodes, into P.kblocks (host sets
// kblocks / kb_stride before the call; eb_mode 2: row_info / ovf_off / ovf_rec as well)
void edge_blocks_build(const PersistParams &P, int32_t B, double *kblocks, int eb_mode, hipStream_t s);
int persist_stamp_words();   // words per workgroup in PersistParams::stamps
bool persist_stamps_built(); // the library was compiled with -DMAG_PERSIST_STAMPS
void mark_published(const int32_t *halo_g, int64_t halo_total, uint8_t *maskP, hipStream_t s);
// bit 3: read through memory by a workgroup of the on-chip kernel (k tiles per workgroup) other than the owner's
// all_rows: sibling tiles read each other's LDS slots whatever their rows' length (eb_mode 2: every entry is remapped at start-up)
void mark_external(const int32_t *halo_g, const TileMeta *meta, int32_t t0, int32_t t1, int32_t B, int32_t k, uint8_t *maskP,
                   bool all_rows, hipStream_t s);

// ---- fp32 leg of BASELINE config 5 (fp64 vs fp32 CG tolerance sweep): the fused iteration with the CG state, the
// operator arithmetic and TILE-RELATIVE coordinates in fp32; dot products accumulate in fp64 ----
struct Rqp32 {
    float2 r, q, p;
};
struct Fused32Params {
    int64_t N;
    int32_t T, nPart, cap, par, hist_len, pad;
    // multi-GPU (all zero / null on one GPU): tile and node range of this rank, the exchange buffers of the streaming
    // protocol (k_cg_fused's, with q converted to fp64 on the way: one all-reduce of doubles serves both precisions)
    int32_t t0, t1, own0, own1, n_iface, part_stride_in;
    const int32_t *iface, *own_qslot, *halo_qslot;
    const double2 *comm_in_q;
    double2 *comm_out_q;
    const float2 *xyP32;     // owned nodes, relative to their tile's first node
    const uint8_t *maskP;
    const TileMeta *meta;
    const uint32_t *ell16;
    const int32_t *halo_g;
    const float2 *halo_xy32; // halo nodes, relative to the READING tile's first node
    float c0, nu, h;
    const Rqp32 *in;
    Rqp32 *out;
    float2 *x;
    const double *part_in;
    double *part_out;
    int32_t part_stride;
    int32_t pad2;
    FusedState *st;
    double *hist;
};
void coords32(const double *xyP, const int32_t *halo_g, const int32_t *tile_hoff, int64_t N, int32_t B, int32_t T,
              float *xyP32, float *halo_xy32, hipStream_t s);
void fused32_launch(const Fused32Params &P, int32_t B, int32_t grid, hipStream_t s);
int fused32_grid(int32_t B, int32_t cap, int32_t tiles);
void fused32_init(const double2 *bP, Rqp32 *in, Rqp32 *out, float2 *x, int64_t N, int32_t B, int32_t T, int32_t t0,
                  int32_t t1, double *part, int32_t stride, int32_t grid, hipStream_t s);
void x32_to_f64(const float2 *x32, int64_t N, double2 *x, hipStream_t s);

int cg_grid(int32_t T);
// operator kernel, B in {256,512,1024}; cg_mode: p = -r + beta*pprev fused, writes pnew, q, partPQ
void op_launch(const OpParams &P, int32_t B, bool cg_mode, hipStream_t s);
void upd_launch(const UpdParams &P, int32_t B, hipStream_t s);
// r = -bP, x = 0 handled by caller memset; writes partRR
void cg_init(const double2 *bP, double2 *r, int64_t N, int32_t B, int32_t T, int32_t t0, int32_t t1, double *partRR,
             hipStream_t s);
// multi-GPU: one buffer per all-reduce, [dot partial sum | interface values]
void iface_pack(const double *part, int nPart, const double2 *v, const int32_t *iface, int32_t n_iface, int32_t own0,
                int32_t own1, double *buf, hipStream_t s);
void iface_unpack(const double *buf, const int32_t *iface, int32_t n_iface, int32_t own0, int32_t own1, double2 *v,
                  hipStream_t s);
// one block: bb = sum(partRR), thresholds, counters
void cg_setup(const double *partRR, int32_t nPart, int stop_mode, double tol, long long max_iter, CgState *st,
              hipStream_t s);
// vP[2*iperm[i]+a] = (mask? !known : 1) * v[2i+a]  /  y[2i+a] = yP[2*iperm[i]+a]
void to_hilbert(const double *v, const int32_t *iperm, const uint8_t *u_known, int32_t masked, int64_t N, double *vP,
                hipStream_t s);
void from_hilbert(const double *yP, const int32_t *iperm, int64_t N, double *y, hipStream_t s);
// uext[2*iperm[i]+a] = known ? u_in : 0
void known_to_hilbert(const double *u_in, const uint8_t *u_known, const int32_t *iperm, int64_t N, double *uP,
                      hipStream_t s);
// bP = free ? f_in - yP : 0 (Hilbert order)
void rhs_from_apply(const double *yP, const double *f_in, const uint8_t *u_known, const uint32_t *perm, int64_t N,
                    double *bP, hipStream_t s);

} // namespace magk
