// Integer kernels of the one-time symbolic phases (no reference counterpart:
// the reference scatters into a dense n x n matrix, solver.rs:295-325).
//   * Hilbert ordering of nodes  -> contiguous runs of B nodes are compact tiles
//   * node -> incident (element, corner) lists, ascending element order
//   * per-tile ELL table of the other two corners of every incident element
//   * node-block CSR pattern of K in the caller's numbering, ascending columns
#include <cstring>

#include <type_traits>

#include <hip/hip_runtime.h>

#include "kernels.h"

namespace magk {

static inline int blocks_for(int64_t n, int threads, int cap = 1 << 30)
{
    int64_t b = (n + threads - 1) / threads;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// ---------------------------------------------------------------- bbox ---
__device__ inline void wave_minmax(double &lo, double &hi)
{
    for (int off = 32; off > 0; off >>= 1) {
        lo = fmin(lo, __shfl_down(lo, off));
        hi = fmax(hi, __shfl_down(hi, off));
    }
}

__global__ void __launch_bounds__(256) k_bbox_partial(const double2 *xy, int64_t N, double *part)
{
    __shared__ double s[4][4];
    double xlo = 1.0e308, ylo = 1.0e308, xhi = -1.0e308, yhi = -1.0e308;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const double2 c = xy[i];
        xlo = fmin(xlo, c.x);
        xhi = fmax(xhi, c.x);
        ylo = fmin(ylo, c.y);
        yhi = fmax(yhi, c.y);
    }
    wave_minmax(xlo, xhi);
    wave_minmax(ylo, yhi);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s[w][0] = xlo;
        s[w][1] = ylo;
        s[w][2] = xhi;
        s[w][3] = yhi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            s[0][0] = fmin(s[0][0], s[k][0]);
            s[0][1] = fmin(s[0][1], s[k][1]);
            s[0][2] = fmax(s[0][2], s[k][2]);
            s[0][3] = fmax(s[0][3], s[k][3]);
        }
        for (int k = 0; k < 4; ++k) part[4 * blockIdx.x + k] = s[0][k];
    }
}

// one wave: lane k takes partials k, k + 64, ... (min and max are exact and order-independent: the same box whatever the
// order), then the wave tree of the partial kernel.  (A single thread walking the partials took 40 us.)
__global__ void __launch_bounds__(64) k_bbox_final(const double *part, int nblocks, double *bbox4)
{
    double xlo = 1.0e308, ylo = 1.0e308, xhi = -1.0e308, yhi = -1.0e308;
    for (int k = threadIdx.x; k < nblocks; k += 64) {
        xlo = fmin(xlo, part[4 * k]);
        ylo = fmin(ylo, part[4 * k + 1]);
        xhi = fmax(xhi, part[4 * k + 2]);
        yhi = fmax(yhi, part[4 * k + 3]);
    }
    wave_minmax(xlo, xhi);
    wave_minmax(ylo, yhi);
    if (threadIdx.x == 0) {
        bbox4[0] = xlo;
        bbox4[1] = ylo;
        bbox4[2] = xhi;
        bbox4[3] = yhi;
    }
}

void bbox(const double *xy, int64_t N, double *scratch, double *bbox4, hipStream_t s)
{
    const int nb = blocks_for(N, 256, 256);
    k_bbox_partial<<<nb, 256, 0, s>>>((const double2 *)xy, N, scratch);
    k_bbox_final<<<1, 64, 0, s>>>(scratch, nb, bbox4);
}

// ------------------------------------------------------------- Hilbert ---
__device__ inline uint32_t hilbert16(uint32_t x, uint32_t y)
{
    uint32_t d = 0;
    for (uint32_t s = 1u << (kHilbertBits - 1); s > 0; s >>= 1) {
        const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d += s * s * ((3u * rx) ^ ry);
        if (ry == 0) {
            if (rx == 1) {
                x = ((1u << kHilbertBits) - 1u) - x;
                y = ((1u << kHilbertBits) - 1u) - y;
            }
            const uint32_t t = x;
            x = y;
            y = t;
        }
    }
    return d;
}

__global__ void __launch_bounds__(256) k_hilbert_keys(const double2 *xy, int64_t N, const double *bbox4,
                                                      uint32_t *keys, uint32_t *ids)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double x0 = bbox4[0], y0 = bbox4[1];
    const double ext = fmax(bbox4[2] - x0, bbox4[3] - y0);
    const double sc = ext > 0.0 ? 65535.0 / ext : 0.0; // one scale for both axes: tiles stay square
    const double2 c = xy[i];
    double fx = (c.x - x0) * sc, fy = (c.y - y0) * sc;
    // NaN/inf coordinates must not index out of range
    fx = (fx >= 0.0 && fx <= 65535.0) ? fx : 0.0;
    fy = (fy >= 0.0 && fy <= 65535.0) ? fy : 0.0;
    keys[i] = hilbert16((uint32_t)fx, (uint32_t)fy);
    ids[i] = (uint32_t)i;
}

void hilbert_keys(const double *xy, int64_t N, const double *bbox4, uint32_t *keys, uint32_t *ids, hipStream_t s)
{
    k_hilbert_keys<<<blocks_for(N, 256), 256, 0, s>>>((const double2 *)xy, N, bbox4, keys, ids);
}

// cdeg (may be null): triangles per node in caller numbering (k_count_degree) -> deg in the new numbering, so that
// k_incidence_keys need not count them again with atomics.
// f_in / bP (may be null): the right-hand side of a row without a prescribed column, b = 0.0 + f (0 on a prescribed DOF;
// solver.rs:427-432 with nothing known next to the row), written here in the new order while the node's other data are
// gathered anyway; the rows that do have a prescribed column are redone from K later (exact.hip, k_rhs_touched).
__global__ void __launch_bounds__(256) k_apply_order(const uint32_t *perm, const double2 *xy, const uint8_t *u_known,
                                                     int64_t N, int32_t *iperm, double2 *xyP, uint8_t *maskP,
                                                     int32_t *known_count, const int32_t *cdeg, int32_t *deg,
                                                     const double2 *f_in, double2 *bP)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t o = perm[i];
    iperm[o] = (int32_t)i;
    if (cdeg) {
        deg[i] = cdeg[o];
        if (i == 0) deg[N] = 0;
    }
    xyP[i] = xy[o];
    const int kx = u_known[2 * (int64_t)o] ? 1 : 0, ky = u_known[2 * (int64_t)o + 1] ? 1 : 0;
    maskP[i] = (uint8_t)(kx | (ky << 1));
    if (bP) {
        const double2 f = f_in[o];
        bP[i] = make_double2(kx ? 0.0 : 0.0 + f.x, ky ? 0.0 : 0.0 + f.y);
    }
    if (kx + ky) atomicAdd(known_count, kx + ky); // prescribed displacements (solver.rs:370)
}

void apply_order(const uint32_t *perm, const double *xy, const uint8_t *u_known, int64_t N, int32_t *iperm,
                 double *xyP, uint8_t *maskP, int32_t *known_count, const int32_t *cdeg, int32_t *deg, const double *f_in,
                 double *bP, hipStream_t s)
{
    k_apply_order<<<blocks_for(N, 256), 256, 0, s>>>(perm, (const double2 *)xy, u_known, N, iperm, (double2 *)xyP,
                                                     maskP, known_count, cdeg, deg, (const double2 *)f_in, (double2 *)bP);
}

// ---- within-tile order by valence (round 4) ----
// The tiles are runs of B nodes of the Hilbert order; INSIDE a tile the order is free.  The on-chip CG kernel keeps six edge
// blocks per node in registers and a node's further blocks in an LDS pool, and a wave pays for the longest row among its
// 64 lanes -- with the plain Hilbert order nearly every wave of a gmsh-type mesh holds a node of valence 7 (a quarter of
// the nodes) and most hold one of valence 8.  So each tile is stably partitioned by class = min(7, max(0, triangles - 6)):
// the nodes with six triangles or fewer first, in their Hilbert order, then those with seven, eight, ...; the result is then
// rotated by a tile-dependent quarter (below), so that the long rows of a workgroup's four tiles fall on different waves.  A tile
// whose nodes all have class 0 -- every tile of a structured mesh -- keeps its order: the permutation is the identity there.
// cdeg: triangles per node in CALLER numbering (k_count_degree).  One workgroup of B threads per tile.
// (A block takes 4096 consecutive elements.  When their node ids lie within kCountBins of each other -- meshes whose elements
// and nodes are numbered along the geometry, as mesh generators leave them -- the block counts in LDS and adds every node's
// count to memory once: a third to a sixth of the atomics; otherwise one atomic per corner, as before.)
constexpr int kCountPer = 48, kCountBins = 12288;
__global__ void __launch_bounds__(256) k_count_degree(const int32_t *conn, int64_t n3, int64_t N, int32_t *cdeg)
{
    __shared__ int32_t s_lo, s_hi;
    __shared__ uint32_t s_h[kCountBins];
    const int64_t base = (int64_t)blockIdx.x * 256 * kCountPer;
    if (threadIdx.x == 0) {
        s_lo = 0x7fffffff;
        s_hi = -1;
    }
    __syncthreads();
    int32_t lo = 0x7fffffff, hi = -1;
    for (int j = 0; j < kCountPer; ++j) {
        const int64_t k = base + (int64_t)j * 256 + threadIdx.x;
        if (k >= n3) break;
        const int32_t n = conn[k];
        if (n >= 0 && (int64_t)n < N) { // (an index out of range is reported by the incidence kernel)
            lo = n < lo ? n : lo;
            hi = n > hi ? n : hi;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off));
        hi = max(hi, __shfl_xor(hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&s_lo, lo);
        atomicMax(&s_hi, hi);
    }
    __syncthreads();
    lo = s_lo;
    hi = s_hi;
    if (hi < lo) return;
    const int32_t range = hi - lo + 1;
    const bool local = range <= kCountBins;
    if (local) {
        for (int b = threadIdx.x; b < range; b += 256) s_h[b] = 0;
        __syncthreads();
    }
    for (int j = 0; j < kCountPer; ++j) {
        const int64_t k = base + (int64_t)j * 256 + threadIdx.x;
        if (k >= n3) break;
        const int32_t n = conn[k];
        if (n < 0 || (int64_t)n >= N) continue;
        if (local)
            atomicAdd(&s_h[n - lo], 1u);
        else
            atomicAdd(&cdeg[n], 1);
    }
    if (!local) return;
    __syncthreads();
    for (int b = threadIdx.x; b < range; b += 256) {
        const uint32_t c = s_h[b];
        if (c) atomicAdd(&cdeg[lo + b], (int32_t)c);
    }
}

template <int B>
__global__ void __launch_bounds__(B) k_tile_valence_partition(const uint32_t *perm_in, const int32_t *cdeg, int64_t N,
                                                              uint32_t *perm_out)
{
    constexpr int NW = B / 64, NC = 8;
    __shared__ int32_t s_cnt[NC][NW];
    const int32_t t = blockIdx.x, l = threadIdx.x, wave = l >> 6;
    const int64_t i = (int64_t)t * B + l;
    const bool valid = i < N;
    const uint32_t o = valid ? perm_in[i] : 0u;
    int32_t c = 0;
    if (valid) {
        const int32_t d = cdeg[o] - 6;
        c = d < 0 ? 0 : (d > NC - 1 ? NC - 1 : d);
    }
    int32_t rank_in_wave = 0;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const unsigned long long m = __ballot(valid && c == k);
        if (c == k) rank_in_wave = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if ((l & 63) == 0) s_cnt[k][wave] = __popcll(m);
    }
    __syncthreads();
    int32_t base = 0;
    for (int k = 0; k < NC; ++k)
        for (int w = 0; w < NW; ++w)
            if (k < c || (k == c && w < wave)) base += s_cnt[k][w];
    // ... and the partitioned order is ROTATED by a quarter of the tile per tile index: the long rows at its end fall on waves
    // 6-7, 0-1, 2-3, 4-5 for the four tiles a workgroup of the on-chip kernel holds -- every wave gets one tile's long rows.
    // (a whole tile only: the last, partial tile keeps the plain partition; a tile without long rows is rotated all the same
    // unless EVERY class is 0 -- the structured meshes' tiles stay as the Hilbert order left them)
    int32_t pos = base + rank_in_wave;
    const bool whole = (int64_t)(t + 1) * B <= N;
    int32_t nz = 0;
    for (int w = 0; w < NW; ++w) nz += s_cnt[0][w];
    if (whole && nz != B) pos = (pos + (B / 4) * (t & 3)) & (B - 1);
    if (valid) perm_out[(int64_t)t * B + pos] = o;
}

void count_degree(const int32_t *conn, int64_t E, int64_t N, int32_t *cdeg, hipStream_t s)
{
    k_count_degree<<<blocks_for(3 * E, 256 * kCountPer), 256, 0, s>>>(conn, 3 * E, N, cdeg);
}

void tile_valence_partition(const uint32_t *perm_in, const int32_t *cdeg, int64_t N, int32_t B, int32_t T, uint32_t *perm_out,
                            hipStream_t s)
{
    if (B == 256)
        k_tile_valence_partition<256><<<T, 256, 0, s>>>(perm_in, cdeg, N, perm_out);
    else if (B == 512)
        k_tile_valence_partition<512><<<T, 512, 0, s>>>(perm_in, cdeg, N, perm_out);
    else
        (void)hipMemcpyAsync(perm_out, perm_in, 4 * (size_t)N, hipMemcpyDeviceToDevice, s);
}

// ----------------------------------------------------------- incidence ---
__global__ void __launch_bounds__(256) k_incidence_keys(const int32_t *conn, int64_t n3, const int32_t *iperm,
                                                        int64_t N, uint32_t *keys, uint32_t *vals, int32_t *deg,
                                                        int32_t *err)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n3) return;
    int32_t n = conn[k];
    if (n < 0 || (int64_t)n >= N) {
        atomicOr(err, 1);
        n = 0;
    }
    const int32_t g = iperm[n];
    keys[k] = (uint32_t)g;
    vals[k] = (uint32_t)k;
    if (deg) atomicAdd(&deg[g], 1); // (null: already counted in caller numbering and carried over by k_apply_order)
}

void incidence_keys(const int32_t *conn, int64_t E, const int32_t *iperm, int64_t N, uint32_t *keys, uint32_t *vals,
                    int32_t *deg, int32_t *err, hipStream_t s)
{
    k_incidence_keys<<<blocks_for(3 * E, 256), 256, 0, s>>>(conn, 3 * E, iperm, N, keys, vals, deg, err);
}

__global__ void __launch_bounds__(256) k_tile_degree(const int32_t *deg, int64_t N, int32_t B, int32_t *tile_deg,
                                                     int64_t *tile_cnt)
{
    __shared__ int s[4];
    const int64_t base = (int64_t)blockIdx.x * B;
    int m = 0;
    for (int l = threadIdx.x; l < B; l += 256) {
        const int64_t i = base + l;
        if (i < N) m = max(m, deg[i]);
    }
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off));
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(s[0], s[1]), max(s[2], s[3]));
        tile_deg[blockIdx.x] = m;
        tile_cnt[blockIdx.x] = (int64_t)m * B;
    }
}

void tile_degree(const int32_t *deg, int64_t N, int32_t B, int32_t T, int32_t *tile_deg, int64_t *tile_cnt,
                 hipStream_t s)
{
    k_tile_degree<<<T, 256, 0, s>>>(deg, N, B, tile_deg, tile_cnt);
}

__global__ void __launch_bounds__(256) k_fill_ell(const int32_t *inc_off, const uint32_t *inc, const int32_t *conn,
                                                  const int32_t *iperm, const int32_t *tile_deg,
                                                  const int64_t *tile_off, int64_t N, int32_t B, int2 *ell)
{
    const int32_t t = blockIdx.x;
    const int32_t td = tile_deg[t];
    int2 *dst = ell + tile_off[t];
    for (int l = threadIdx.x; l < B; l += 256) {
        const int64_t i = (int64_t)t * B + l;
        int32_t o = 0, d = 0;
        if (i < N) {
            o = inc_off[i];
            d = inc_off[i + 1] - o;
        }
        for (int k = 0; k < td; ++k) {
            int2 bc = make_int2(-1, -1);
            if (k < d) {
                const uint32_t v = inc[o + k];
                const uint32_t e = v / 3u, c = v - 3u * e;
                const uint32_t c1 = c == 2 ? 0 : c + 1, c2 = c1 == 2 ? 0 : c1 + 1;
                bc.x = iperm[conn[3 * (int64_t)e + c1]];
                bc.y = iperm[conn[3 * (int64_t)e + c2]];
            }
            dst[(int64_t)k * B + l] = bc;
        }
    }
}

void fill_ell(const int32_t *inc_off, const uint32_t *inc, const int32_t *conn, const int32_t *iperm,
              const int32_t *tile_deg, const int64_t *tile_off, int64_t N, int32_t B, int32_t T, int2 *ell,
              hipStream_t s)
{
    k_fill_ell<<<T, 256, 0, s>>>(inc_off, inc, conn, iperm, tile_deg, tile_off, N, B, ell);
}

// ------------------------------------------ tile-local numbering (halo) ---
// A tile's halo = distinct nodes referenced by its nodes' incident elements but owned by another tile.
// count / emit (tile<<32 | node) refs per node, sort + unique (host side drives rocPRIM), then the ELL
// table is rewritten with 16-bit tile-local ids: owned l in [0,B), halo B + rank in the tile's sorted list.
template <bool EMIT>
__global__ void __launch_bounds__(256) k_halo_refs(const int32_t *inc_off, const uint32_t *inc, const int32_t *conn,
                                                   const int32_t *iperm, int64_t N, int32_t B, int32_t *cnt,
                                                   const int32_t *off, uint64_t *keys)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > N) return;
    int32_t c = 0;
    if (i < N) {
        const int32_t t = (int32_t)(i / B);
        const int32_t o = inc_off[i], d = inc_off[i + 1] - o;
        int64_t w = EMIT ? off[i] : 0;
        for (int k = 0; k < d; ++k) {
            const uint32_t v = inc[o + k];
            const uint32_t e = v / 3u, cc = v - 3u * e;
            const uint32_t c1 = cc == 2 ? 0 : cc + 1, c2 = c1 == 2 ? 0 : c1 + 1;
            const int32_t g1 = iperm[conn[3 * (int64_t)e + c1]], g2 = iperm[conn[3 * (int64_t)e + c2]];
            if (g1 / B != t) {
                if (EMIT) keys[w++] = ((uint64_t)(uint32_t)t << 32) | (uint32_t)g1;
                ++c;
            }
            if (g2 / B != t) {
                if (EMIT) keys[w++] = ((uint64_t)(uint32_t)t << 32) | (uint32_t)g2;
                ++c;
            }
        }
    }
    if (!EMIT) cnt[i] = c; // entry N is the scan's sentinel
}

void halo_count(const int32_t *inc_off, const uint32_t *inc, const int32_t *conn, const int32_t *iperm, int64_t N,
                int32_t B, int32_t *cnt, hipStream_t s)
{
    k_halo_refs<false><<<blocks_for(N + 1, 256), 256, 0, s>>>(inc_off, inc, conn, iperm, N, B, cnt, nullptr, nullptr);
}

void halo_emit(const int32_t *inc_off, const uint32_t *inc, const int32_t *conn, const int32_t *iperm, int64_t N,
               int32_t B, const int32_t *off, uint64_t *keys, hipStream_t s)
{
    k_halo_refs<true><<<blocks_for(N + 1, 256), 256, 0, s>>>(inc_off, inc, conn, iperm, N, B, nullptr, off, keys);
}

__global__ void __launch_bounds__(256) k_halo_unique(const uint64_t *keys, const int32_t *head, const int32_t *blk,
                                                     int64_t n, int32_t *halo_g, int32_t *tile_hcnt)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n || !head[k]) return;
    const uint64_t key = keys[k];
    halo_g[blk[k]] = (int32_t)(key & 0xffffffffu);
    atomicAdd(&tile_hcnt[(int32_t)(key >> 32)], 1);
}

void halo_unique(const uint64_t *keys, const int32_t *head, const int32_t *blk, int64_t n, int32_t *halo_g,
                 int32_t *tile_hcnt, hipStream_t s)
{
    k_halo_unique<<<blocks_for(n, 256), 256, 0, s>>>(keys, head, blk, n, halo_g, tile_hcnt);
}

__global__ void __launch_bounds__(256) k_halo_coords(const int32_t *halo_g, const double2 *xyP, int64_t n,
                                                     double2 *halo_xy)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) halo_xy[i] = xyP[halo_g[i]];
}

void halo_coords(const int32_t *halo_g, const double *xyP, int64_t n, double *halo_xy, hipStream_t s)
{
    if (n > 0) k_halo_coords<<<blocks_for(n, 256), 256, 0, s>>>(halo_g, (const double2 *)xyP, n, (double2 *)halo_xy);
}

__device__ inline uint32_t local_id(int32_t g, int32_t base, int32_t B, const int32_t *hl, int32_t nh)
{
    const uint32_t l = (uint32_t)(g - base);
    if (l < (uint32_t)B) return l;
    int32_t lo = 0, hi = nh - 1;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (hl[mid] < g)
            lo = mid + 1;
        else
            hi = mid;
    }
    return (uint32_t)(B + lo);
}

__global__ void __launch_bounds__(256) k_fill_ell16(const int32_t *inc_off, const uint32_t *inc, const int32_t *conn,
                                                    const int32_t *iperm, const int32_t *tile_deg,
                                                    const int64_t *tile_off, const int32_t *tile_hoff,
                                                    const int32_t *halo_g, int64_t N, int32_t B, uint32_t *ell,
                                                    uint32_t *ell_asm, uint16_t *ell_pos)
{
    const int32_t t = blockIdx.x;
    const int32_t td = tile_deg[t];
    if (td == 0) return; // no triangle at any node of the tile (sharded ordering phase: a tile this rank does not need)
    const int32_t base = t * B;
    const int32_t *hl = halo_g + tile_hoff[t];
    const int32_t nh = tile_hoff[t + 1] - tile_hoff[t];
    uint32_t *dst = ell + tile_off[t];
    // the assembly's copy (k_assemble_fan, exact.hip): the same ids with the node's corner label and the pairing of the
    // node's triangles in the spare bits; it stays as written here, while `ell` is rewritten into ring form by k_ring16
    uint32_t *dst_asm = ell_asm ? ell_asm + tile_off[t] : nullptr;
    uint16_t *dst_pos = ell_asm ? ell_pos + tile_off[t] : nullptr;
    for (int l = threadIdx.x; l < B; l += 256) {
        const int64_t i = (int64_t)base + l;
        int32_t o = 0, d = 0;
        if (i < N) {
            o = inc_off[i];
            d = inc_off[i + 1] - o;
        }
        for (int k = 0; k < td; ++k) {
            uint32_t w = 0xffffffffu;
            if (k < d) {
                const uint32_t v = inc[o + k];
                const uint32_t e = v / 3u, c = v - 3u * e;
                const uint32_t c1 = c == 2 ? 0 : c + 1, c2 = c1 == 2 ? 0 : c1 + 1;
                const uint32_t lb = local_id(iperm[conn[3 * (int64_t)e + c1]], base, B, hl, nh);
                const uint32_t lc = local_id(iperm[conn[3 * (int64_t)e + c2]], base, B, hl, nh);
                w = lb | (lc << 16);
            }
            dst[(int64_t)k * B + l] = w;
        }
        if (!dst_asm) continue;
        // The assembly's words: lb | label << 12 | flags << 14 | lc << 16 | partner << 28 | has_partner << 31, where the
        // PARTNER of triangle k is the triangle of the same node whose c is k's b (the other side of the edge a-b) and
        // flag bit 0 = "k's c is nobody's b" (open end of a fan), bit 1 = the node cannot be assembled triangle by
        // triangle (an element listing a node twice, an edge with three or more triangles, more than eight triangles).
        // Second pass over the row just written (ids back from `dst`: still in pre-ring form, this thread's own stores).
        // (rows of up to eight triangles are held in registers for this; longer ones are flagged and never paired)
        uint32_t ws[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) ws[k] = (k < d && k < td) ? dst[(int64_t)k * B + l] : 0xffffffffu;
        bool odd = d > 8;
        uint32_t pjs = 0, has = 0, openc = 0; // per triangle: partner (3 bits each), has a partner, its c is nobody's b
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (ws[k] == 0xffffffffu) continue;
            const uint32_t lbk = ws[k] & 0xffffu, lck = ws[k] >> 16;
            int npb = 0, npc = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (ws[j] == 0xffffffffu) continue;
                if ((ws[j] >> 16) == lbk) {
                    ++npb;
                    pjs = (pjs & ~(7u << (3 * k))) | ((uint32_t)j << (3 * k));
                }
                npc += (ws[j] & 0xffffu) == lck ? 1 : 0;
            }
            has |= (npb > 0 ? 1u : 0u) << k;
            openc |= (npc == 0 ? 1u : 0u) << k;
            odd |= lbk == lck || lbk == (uint32_t)l || lck == (uint32_t)l || npb > 1 || npc > 1;
        }
        // Where the blocks go: the row's columns are the node and its neighbours in ascending CALLER id (k_pattern_rows), and
        // in a fan every neighbour is the b of exactly one triangle or the c of an open end, so the position of a column is a
        // count over those: kb (bits 0-3) for triangle k's b, kc (4-7) for its c, kd (8-11, in every word of the node) for
        // the diagonal.  At most eight triangles, hence at most ten columns.
        int32_t cb[8], cc[8], self = 0;
        uint32_t lab[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            cb[k] = cc[k] = 0x7fffffff;
            lab[k] = 0;
            if (ws[k] == 0xffffffffu) continue;
            const uint32_t v = inc[o + k];
            const uint32_t e = v / 3u, c = v - 3u * e;
            const uint32_t c1 = c == 2 ? 0 : c + 1, c2 = c1 == 2 ? 0 : c1 + 1;
            lab[k] = c;
            self = conn[3 * (int64_t)e + c];
            cb[k] = conn[3 * (int64_t)e + c1];
            cc[k] = ((openc >> k) & 1u) ? conn[3 * (int64_t)e + c2] : 0x7fffffff; // counted only where it is nobody's b
        }
        auto before = [&](int32_t x) { // columns of the row with a caller id below x
            int n = self < x ? 1 : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) n += (cb[j] < x ? 1 : 0) + (cc[j] < x ? 1 : 0);
            return (uint32_t)n;
        };
        const uint32_t kd = before(self);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k >= td) break;
            uint32_t wa = 0xffffffffu, wp = kd << 8;
            if (ws[k] != 0xffffffffu) {
                wa = (ws[k] & 0xffffu) | (lab[k] << 12) | (((openc >> k) & 1u) << 14) | ((odd ? 1u : 0u) << 15) |
                     ((ws[k] >> 16) << 16) | (((pjs >> (3 * k)) & 7u) << 28) | (((has >> k) & 1u) << 31);
                wp |= (before(cb[k]) & 15u) | ((((openc >> k) & 1u) ? before(cc[k]) & 15u : 0u) << 4);
            }
            dst_asm[(int64_t)k * B + l] = wa;
            dst_pos[(int64_t)k * B + l] = (uint16_t)wp;
        }
        for (int k = 8; k < td; ++k) { // a node of more than eight triangles is flagged (odd): only ids and label matter
            uint32_t wa = 0xffffffffu;
            if (k < d) {
                const uint32_t v = inc[o + k];
                const uint32_t wk = dst[(int64_t)k * B + l];
                wa = (wk & 0xffffu) | ((v - 3u * (v / 3u)) << 12) | (1u << 15) | ((wk >> 16) << 16);
            }
            dst_asm[(int64_t)k * B + l] = wa;
            dst_pos[(int64_t)k * B + l] = 0;
        }
    }
}

void fill_ell16(const int32_t *inc_off, const uint32_t *inc, const int32_t *conn, const int32_t *iperm,
                const int32_t *tile_deg, const int64_t *tile_off, const int32_t *tile_hoff, const int32_t *halo_g,
                int64_t N, int32_t B, int32_t T, uint32_t *ell, uint32_t *ell_asm, uint16_t *ell_pos, hipStream_t s)
{
    k_fill_ell16<<<T, 256, 0, s>>>(inc_off, inc, conn, iperm, tile_deg, tile_off, tile_hoff, halo_g, N, B, ell, ell_asm,
                                   ell_pos);
}

// Ring form of the tile-local table.  fill_ell16 leaves one word (lb | lc << 16) per incident triangle (a, b, c);
// around a node consecutive triangles share a neighbour, so the same information is a walk over the neighbours:
// a row becomes a sequence of 16-bit ENTRIES -- tile-local id in bits 0-10, bit 15 = "no triangle between the previous
// entry and this one" (first entry of a fan), 0xffff = end -- and every consecutive pair (prev, cur) without the break
// bit is one triangle (a, prev, cur) in its original orientation.  A closed fan of d triangles takes d + 1 entries
// instead of 2 d ids: half the table bytes per CG iteration and half the LDS gathers in the operator kernels.
// Rewritten in place (entries <= 2 d, two per word); rows of more than kRingMaxDeg triangles just get the break bit
// on every b (each triangle its own fan).  tile_rdeg[t] = words the longest row of tile t now uses.
constexpr int kRingMaxDeg = 32;

__global__ void __launch_bounds__(256) k_ring16(const int32_t *tile_deg, const int64_t *tile_off, int32_t B, uint32_t *ell,
                                                int32_t *tile_rdeg, int32_t nb, uint8_t *row_info)
{
    const int32_t t = blockIdx.x;
    const int32_t td = tile_deg[t];
    if (td == 0) return; // an empty tile: row_info stays at the launcher's 0x80 ("one fan, no entries") for all of its rows
    uint32_t *dst = ell + tile_off[t];
    int32_t wmax = 0, emax = 0;
    // Is every row ONE fan of at most nb entries, or nb + 1 with the last entry closing onto the first?  Then the on-chip CG
    // kernel can fold the mesh into nb edge blocks per node (persist.hip, EB instantiation); one row that is not -- several
    // fans at a node, a closed fan of valence > nb, an open one of > nb - 1 triangles -- and the whole mesh keeps the
    // triangle walk.  tile_rdeg[2 T] collects the answer (0: every row qualifies).
    // Round 4: bit 0 of that word as before ("some row is not a plain fan of <= nb entries"), bit 1 = some row is not ONE fan
    // at all (several fans at a node, more than kRingMaxDeg triangles).  A mesh with bit 1 clear can still run the edge-block
    // kernel: rows with more than nb blocks keep the blocks beyond nb in LDS (persist.hip, EBM == 2), which is what gmsh-type
    // meshes need -- a quarter of their nodes have seven neighbours.  row_info[node] says what that needs per row: bits 0-5 the
    // row's entries n, bit 6 = the fan is closed (entry n - 1 repeats entry 0), bit 7 = the row is one fan.
    bool plain = true, single = true;
    for (int l = threadIdx.x; l < B; l += 256) {
        uint32_t *row = dst + l; // row[k * B], k < td
        int d = 0;
        while (d < td && row[(int64_t)d * B] != 0xffffffffu) ++d;
        int words = 0;
        if (d > kRingMaxDeg) {
            for (int k = 0; k < d; ++k) row[(int64_t)k * B] |= 0x8000u;
            const uint32_t pad = ((row[(int64_t)(d - 1) * B] >> 16) & 0xfffu) | 0x8000u;
            for (int k = d; k < td; ++k) row[(int64_t)k * B] = pad | (pad << 16);
            words = d;
            emax = 2 * d > emax ? 2 * d : emax;
            plain = false;
            single = false;
            if (row_info) row_info[(int64_t)t * B + l] = 0;
        } else if (d > 0) {
            // (MAXD: the arrays live in registers and every loop over them is unrolled to their size -- rows of up to eight
            // triangles, all but a few per cent even of an unstructured mesh, take the small instantiation: the kernel went
            // from 132 to ~45 us at 1M triangles)
            auto ring_row = [&](auto maxd_c) {
            constexpr int MAXD = decltype(maxd_c)::value;
            uint32_t pr[MAXD];
            uint16_t out[2 * MAXD];
#pragma unroll
            for (int k = 0; k < MAXD; ++k) pr[k] = k < d ? row[(int64_t)k * B] : 0xffffffffu;
            uint32_t used = 0;
            int n = 0, remaining = d;
            while (remaining > 0) {
                // fan start: the lowest unused triangle whose b is nobody's c; a closed fan starts at its lowest triangle
                int start = -1, first = -1;
                for (int i = 0; i < d && start < 0; ++i) {
                    if ((used >> i) & 1u) continue;
                    if (first < 0) first = i;
                    const uint32_t bi = pr[i] & 0xffffu;
                    bool fed = false;
                    for (int j = 0; j < d; ++j)
                        if (j != i && !((used >> j) & 1u) && (pr[j] >> 16) == bi) {
                            fed = true;
                            break;
                        }
                    if (!fed) start = i;
                }
                if (start < 0) start = first;
                out[n++] = (uint16_t)((pr[start] & 0xffffu) | 0x8000u);
                uint32_t cur = pr[start] >> 16;
                out[n++] = (uint16_t)cur;
                used |= 1u << start;
                --remaining;
                for (;;) {
                    int nx = -1;
                    for (int i = 0; i < d; ++i)
                        if (!((used >> i) & 1u) && (pr[i] & 0xffffu) == cur) {
                            nx = i;
                            break;
                        }
                    if (nx < 0) break;
                    cur = pr[nx] >> 16;
                    out[n++] = (uint16_t)cur;
                    used |= 1u << nx;
                    --remaining;
                }
            }
            words = (n + 1) / 2;
            emax = n > emax ? n : emax;
            bool one = true;
            for (int k = 1; k < n; ++k) one &= !(out[k] & 0x8000u); // a second fan
            plain &= one;
            plain &= n <= nb || (n == nb + 1 && (out[nb] & 0xfffu) == (out[0] & 0xfffu));
            one &= n < 64;
            single &= one;
            if (row_info)
                row_info[(int64_t)t * B + l] = (uint8_t)((one ? 0x80u | (uint32_t)n : 0u) |
                                                         (one && n >= 3 && (out[n - 1] & 0xfffu) == (out[0] & 0xfffu) ? 0x40u : 0u));
            // Padding repeats the last neighbour with the break bit: a walker may treat EVERY entry of the tile's
            // row length as present (no per-entry test), the repeated entry closes no triangle.
            const uint32_t pad = (uint32_t)(out[n - 1] & 0xfffu) | 0x8000u;
            for (int k = 0; k < td; ++k) {
                const uint32_t lo = 2 * k < n ? out[2 * k] : pad, hi = 2 * k + 1 < n ? out[2 * k + 1] : pad;
                row[(int64_t)k * B] = lo | (hi << 16);
            }
            }; // ring_row
            if (d <= 8)
                ring_row(std::integral_constant<int, 8>{});
            else
                ring_row(std::integral_constant<int, kRingMaxDeg>{});
        } else {
            for (int k = 0; k < td; ++k) row[(int64_t)k * B] = 0x80008000u; // no triangles: slot 0, break, throughout
            if (row_info) row_info[(int64_t)t * B + l] = 0x80u; // no entries, no blocks
        }
        wmax = words > wmax ? words : wmax;
    }
    if (wmax > 0) atomicMax(&tile_rdeg[t], wmax);
    if (emax > 0) atomicMax(&tile_rdeg[gridDim.x + t], emax); // second half of the array: entries of the longest row
    if (!plain || !single) atomicOr(&tile_rdeg[2 * gridDim.x], (plain ? 0 : 1) | (single ? 0 : 2));
}

void ring16(const int32_t *tile_deg, const int64_t *tile_off, int32_t B, int32_t T, uint32_t *ell, int32_t *tile_rdeg,
            int32_t block_entries, uint8_t *row_info, hipStream_t s)
{
    // [0, T): words, [T, 2T): entries of the longest row, [2T]: bit 0 some row is not a plain short fan, bit 1 some row is
    // not one fan (see k_ring16)
    (void)hipMemsetAsync(tile_rdeg, 0, 4 * (2 * (size_t)T + 1), s);
    if (row_info) (void)hipMemsetAsync(row_info, 0x80, (size_t)T * (size_t)B, s);
    k_ring16<<<T, 256, 0, s>>>(tile_deg, tile_off, B, ell, tile_rdeg, block_entries, row_info);
}

// Edge blocks beyond the nb a node keeps in registers (persist.hip, EBM == 2): per node of the padded Hilbert order the
// number of blocks its row needs above nb -- n - 1 for a closed fan of n entries (the closing triangle is folded into
// blocks n - 2 and 0), n for an open one.  The exclusive scan of these counts addresses the overflow records.
__global__ void __launch_bounds__(256) k_ovf_counts(const uint8_t *row_info, int64_t npad, int32_t nb, int32_t *cnt)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > npad) return;
    int32_t c = 0;
    if (i < npad) {
        const uint32_t info = row_info[i];
        const int32_t n = (int32_t)(info & 63u), nblk = (info & 0x40u) ? n - 1 : n;
        c = nblk > nb ? nblk - nb : 0;
    }
    cnt[i] = c; // (cnt[npad] = 0: the scan's last entry is the total)
}

void ovf_counts(const uint8_t *row_info, int64_t npad, int32_t nb, int32_t *cnt, hipStream_t s)
{
    k_ovf_counts<<<blocks_for(npad + 1, 256), 256, 0, s>>>(row_info, npad, nb, cnt);
}

// Largest number of overflow records any workgroup of the on-chip kernel (k tiles each, tiles [t0, t1)) has to hold, and
// the largest count of a single node: out[0], out[1] (zeroed by the caller).
__global__ void __launch_bounds__(256) k_ovf_limits(const int32_t *off, int32_t B, int32_t k, int32_t t0, int32_t t1,
                                                    int32_t *out)
{
    const int32_t g = blockIdx.x; // workgroup of the on-chip kernel
    const int64_t n0 = (int64_t)(t0 + g * k) * B;
    int64_t n1 = n0 + (int64_t)k * B;
    if (n1 > (int64_t)t1 * B) n1 = (int64_t)t1 * B;
    int32_t m = 0;
    for (int64_t i = n0 + threadIdx.x; i < n1; i += 256) {
        const int32_t c = off[i + 1] - off[i];
        m = c > m ? c : m;
    }
    if (m > 0) atomicMax(&out[1], m);
    if (threadIdx.x == 0) atomicMax(&out[0], off[n1] - off[n0]);
}

void ovf_limits(const int32_t *off, int32_t B, int32_t k, int32_t t0, int32_t t1, int32_t *out, hipStream_t s)
{
    const int32_t grid = (t1 - t0 + k - 1) / k;
    if (grid > 0) k_ovf_limits<<<grid, 256, 0, s>>>(off, B, k, t0, t1, out);
}

// ------------------------------------------- sharded ordering phase (several ranks) ---
// The node ORDER is global (every rank must agree on every node's index); everything derived from it -- incidence lists,
// halo lists, ELL / ring words, tile tables -- a rank needs only for the tiles that hold a row it keeps: its own tiles, the tiles
// of its ghost nodes (nodes that share an element with an own node: their rows give the right-hand side of the ghost
// recurrences) and the tiles of the prescribed nodes (their rows give the reactions).  need[t] marks those tiles; nodes of
// other tiles get empty incidence lists, so every later kernel finds nothing to do there.
// (the same pass marks the interface: see k_iface_flags below)
__device__ inline int rank_of_tile(const RankTiles &rt, int32_t t)
{
    int r = 0;
    while (r + 1 < rt.R && t >= rt.lo[r + 1]) ++r;
    return r;
}

__global__ void __launch_bounds__(256) k_need_tiles_elems(const int32_t *conn, int64_t E, const int32_t *iperm, int64_t N,
                                                          int32_t B, int32_t t0, int32_t t1, uint8_t *need, RankTiles rt,
                                                          uint32_t *readers)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    int32_t g[3], t[3];
    bool own = false;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int32_t n = conn[3 * e + c];
        if (n < 0 || (int64_t)n >= N) n = 0; // (reported by the incidence kernel)
        g[c] = iperm[n];
        t[c] = g[c] / B;
        own = own || (t[c] >= t0 && t[c] < t1);
    }
    if (own) {
#pragma unroll
        for (int c = 0; c < 3; ++c) need[t[c]] = 1;
    }
    // The interface without any tile table: a node is read by rank r when it shares an element with a node rank r owns (that
    // is what being in the halo of one of r's tiles means): the other corners' owners are OR-ed into a byte per node.
    if (t[0] == t[1] && t[1] == t[2]) return;
    int o[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = rank_of_tile(rt, t[c]);
    if (o[0] == o[1] && o[1] == o[2]) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        uint32_t m = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (o[d] != o[c]) m |= 1u << (o[d] & 7);
        if (m) atomicOr(&readers[g[c] >> 2], m << (8 * (g[c] & 3)));
    }
}

__global__ void __launch_bounds__(256) k_need_tiles_nodes(const uint8_t *maskP, int64_t N, int32_t B, uint8_t *need)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g < N && (maskP[g] & 3)) need[g / B] = 1;
}

void need_tiles(const int32_t *conn, int64_t E, const int32_t *iperm, const uint8_t *maskP, int64_t N, int32_t B, int32_t T,
                int32_t t0, int32_t t1, bool prescribed_rows, uint8_t *need, const RankTiles &rt, uint8_t *readers, hipStream_t s)
{
    (void)hipMemsetAsync(need, 0, (size_t)T, s);
    if (t1 > t0) (void)hipMemsetAsync(need + t0, 1, (size_t)(t1 - t0), s);
    (void)hipMemsetAsync(readers, 0, (((size_t)N + 3) / 4) * 4, s);
    k_need_tiles_elems<<<blocks_for(E, 256), 256, 0, s>>>(conn, E, iperm, N, B, t0, t1, need, rt, (uint32_t *)readers);
    if (prescribed_rows) k_need_tiles_nodes<<<blocks_for(N, 256), 256, 0, s>>>(maskP, N, B, need);
}

// incidence pairs of the needed tiles' nodes only: flag, scan (caller), emit in the order of k -- the stable sort by node then
// leaves every node's elements ascending, exactly as the unsharded list has them
__global__ void __launch_bounds__(256) k_incidence_flags(const int32_t *conn, int64_t n3, const int32_t *iperm, int64_t N,
                                                         int32_t B, const uint8_t *need, int32_t *flag, int32_t *err)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k > n3) return;
    int32_t f = 0;
    if (k < n3) {
        int32_t n = conn[k];
        if (n < 0 || (int64_t)n >= N) {
            atomicOr(err, 1);
            n = 0;
        }
        f = need[iperm[n] / B] ? 1 : 0;
    }
    flag[k] = f; // (flag[n3] = 0: the scan's total lands there)
}

__global__ void __launch_bounds__(256) k_incidence_emit(const int32_t *conn, int64_t n3, const int32_t *iperm, int64_t N,
                                                        const int32_t *off, uint32_t *keys, uint32_t *vals, int32_t *deg)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n3) return;
    const int32_t o = off[k];
    if (off[k + 1] == o) return;
    int32_t n = conn[k];
    if (n < 0 || (int64_t)n >= N) n = 0;
    const int32_t g = iperm[n];
    keys[o] = (uint32_t)g;
    vals[o] = (uint32_t)k;
    if (deg) atomicAdd(&deg[g], 1);
}

void incidence_flags(const int32_t *conn, int64_t E, const int32_t *iperm, int64_t N, int32_t B, const uint8_t *need,
                     int32_t *flag, int32_t *err, hipStream_t s)
{
    k_incidence_flags<<<blocks_for(3 * E + 1, 256), 256, 0, s>>>(conn, 3 * E, iperm, N, B, need, flag, err);
}

void incidence_emit(const int32_t *conn, int64_t E, const int32_t *iperm, int64_t N, const int32_t *off, uint32_t *keys,
                    uint32_t *vals, int32_t *deg, hipStream_t s)
{
    k_incidence_emit<<<blocks_for(3 * E, 256), 256, 0, s>>>(conn, 3 * E, iperm, N, off, keys, vals, deg);
}

__global__ void __launch_bounds__(256) k_zero_unneeded_deg(const uint8_t *need, int64_t N, int32_t B, int32_t *deg)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g < N && !need[g / B]) deg[g] = 0;
}

void zero_unneeded_deg(const uint8_t *need, int64_t N, int32_t B, int32_t *deg, hipStream_t s)
{
    k_zero_unneeded_deg<<<blocks_for(N, 256), 256, 0, s>>>(need, N, B, deg);
}

// the interface list: the compaction, in Hilbert order, of the nodes k_need_tiles_elems has marked -- the same sorted list and the
// same reader masks every rank used to derive from the replicated halo lists
__global__ void __launch_bounds__(256) k_iface_flags(const uint8_t *readers, int64_t N, int32_t *flag)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g <= N) flag[g] = g < N && readers[g] ? 1 : 0;
}

__global__ void __launch_bounds__(256) k_iface_emit(const uint8_t *readers, const int32_t *off, int64_t N, int32_t *iface,
                                                    uint8_t *iface_readers)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= N || !readers[g]) return;
    iface[off[g]] = (int32_t)g;
    iface_readers[off[g]] = readers[g];
}

void iface_flags(const uint8_t *readers, int64_t N, int32_t *flag, hipStream_t s)
{
    k_iface_flags<<<blocks_for(N + 1, 256), 256, 0, s>>>(readers, N, flag);
}

void iface_emit(const uint8_t *readers, const int32_t *off, int64_t N, int32_t *iface, uint8_t *iface_readers, hipStream_t s)
{
    k_iface_emit<<<blocks_for(N, 256), 256, 0, s>>>(readers, off, N, iface, iface_readers);
}

// --------------------------------------------------------- CSR pattern ---
__global__ void __launch_bounds__(256) k_csr_pairs(const int32_t *conn, int64_t n9, uint64_t *keys, uint32_t *vals)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n9) return;
    const int64_t e = k / 9;
    const int ab = (int)(k - 9 * e);
    const int a = ab / 3, b = ab - 3 * a;
    keys[k] = ((uint64_t)(uint32_t)conn[3 * e + a] << 32) | (uint64_t)(uint32_t)conn[3 * e + b];
    vals[k] = (uint32_t)k;
}

void csr_pairs(const int32_t *conn, int64_t E, uint64_t *keys, uint32_t *vals, hipStream_t s)
{
    k_csr_pairs<<<blocks_for(9 * E, 256), 256, 0, s>>>(conn, 9 * E, keys, vals);
}

// ---- CSR pattern straight from the incidence lists (default): no pair list, no 9E-key sort ----
// Row node i's block columns are the distinct nodes of its incident elements (solver.rs:304-322: every (node, node) pair
// of an element gets an entry, zero or not), ascending.  One thread per node of the Hilbert order walks the node's
// incidence entries (already sorted and contiguous: the list the operator tables and the assembly use), gathers the
// elements' node triples four at a time and keeps the distinct ids in a sorted register array of kRowCap slots
// (insertion by compare-and-select: no run-time indexing).  Pass COUNT writes the row's block count (0 for a row this
// rank does not keep), a scan gives bptr, pass FILL writes bcol.  A row with more than kRowCap distinct columns
// (valence >= kRowCap: hub nodes) raises `overflow` and the host falls back to the sort-based pattern (k_csr_pairs ...).
constexpr int kRowCap = 16;

template <bool FILL>
__global__ void __launch_bounds__(256) k_pattern_rows(const int32_t *inc_off, const uint32_t *inc, const uint32_t *perm,
                                                      const int32_t *conn, const uint8_t *local, int64_t N,
                                                      const int32_t *bptr, int32_t *out, int32_t *overflow,
                                                      const uint8_t *u_known, uint8_t *touch, const uint8_t *need, int32_t B)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g > N) return;
    if (g == N) {
        if (!FILL) out[N] = 0; // sentinel for the exclusive scan
        return;
    }
    if (need && !need[g / B]) { // sharded ordering phase: no row of this tile is kept (the counts were zeroed by the launcher)
        if (FILL && touch) touch[g] = 0;
        return;
    }
    const int64_t i = perm[g];
    if (local && !local[i]) {
        if (!FILL) out[i] = 0;
        if (FILL && touch) touch[g] = 0;
        return;
    }
    int32_t cols[kRowCap];
#pragma unroll
    for (int k = 0; k < kRowCap; ++k) cols[k] = 0x7fffffff;
    int n = 0;
    bool over = false;
    auto insert = [&](int32_t v) {
        int p = 0;
        bool dup = false;
#pragma unroll
        for (int k = 0; k < kRowCap; ++k) {
            p += cols[k] < v ? 1 : 0;
            dup |= cols[k] == v;
        }
        if (dup) return;
        if (n == kRowCap) {
            over = true;
            return;
        }
#pragma unroll
        for (int k = kRowCap - 1; k > 0; --k) cols[k] = k > p ? cols[k - 1] : (k == p ? v : cols[k]);
        cols[0] = p == 0 ? v : cols[0];
        ++n;
    };
    const int32_t q1 = inc_off[g + 1];
    for (int32_t q = inc_off[g]; q < q1; q += 4) {
        uint32_t v[4];
        int3 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = q + u < q1 ? inc[q + u] : 0xffffffffu;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (v[u] != 0xffffffffu) t[u] = ((const int3 *)conn)[v[u] / 3u];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (v[u] != 0xffffffffu) {
                insert(t[u].x);
                insert(t[u].y);
                insert(t[u].z);
            }
    }
    if (FILL) {
        const int32_t p = bptr[i];
#pragma unroll
        for (int k = 0; k < kRowCap; ++k)
            if (k < n) out[p + k] = cols[k];
        // While the row's columns are in registers: does the row have a PRESCRIBED column?  One 2-byte gather per block
        // column here instead of four 1-byte gathers per block in the right-hand side's pass over K (exact.hip, rhs_row):
        // rows without one -- all but the boundary's neighbours -- take b = 0.0 + f there and never look at K.
        if (touch) {
            unsigned t = 0;
#pragma unroll
            for (int k = 0; k < kRowCap; ++k)
                if (k < n) t |= ((const uint16_t *)u_known)[cols[k]];
            touch[g] = t ? 1 : 0; // (Hilbert order: the right-hand side's kernel reads one coalesced byte per thread)
        }
    } else {
        out[i] = over ? 0 : n;
        if (over) *overflow = 1;
    }
}

void pattern_count(const int32_t *inc_off, const uint32_t *inc, const uint32_t *perm, const int32_t *conn,
                   const uint8_t *local, int64_t N, int32_t *rowcnt, int32_t *overflow, const uint8_t *need, int32_t B,
                   hipStream_t s)
{
    if (need) (void)hipMemsetAsync(rowcnt, 0, 4 * ((size_t)N + 1), s);
    k_pattern_rows<false><<<blocks_for(N + 1, 256), 256, 0, s>>>(inc_off, inc, perm, conn, local, N, nullptr, rowcnt,
                                                                overflow, nullptr, nullptr, need, B);
}

// touch (N bytes, may be null): 1 for the rows with a prescribed column (u_known: 2N bytes, caller numbering)
void pattern_fill(const int32_t *inc_off, const uint32_t *inc, const uint32_t *perm, const int32_t *conn,
                  const uint8_t *local, int64_t N, const int32_t *bptr, int32_t *bcol, const uint8_t *u_known,
                  uint8_t *touch, const uint8_t *need, int32_t B, hipStream_t s)
{
    k_pattern_rows<true><<<blocks_for(N + 1, 256), 256, 0, s>>>(inc_off, inc, perm, conn, local, N, bptr, bcol, nullptr,
                                                               u_known, touch, need, B);
}

// ---- multi-GPU: the pattern of the rows a rank keeps (owned nodes, one ghost layer, prescribed nodes) ----
// local[i] (caller numbering) = 1 for: nodes of the rank's own Hilbert range, halo nodes of its tiles (their rows give
// the right-hand side of the ghost recurrences), and every node with a prescribed displacement (their rows give the
// reactions, solver.rs:456-469, on every rank without a second collective; they are boundary nodes: O(sqrt N)).
__global__ void __launch_bounds__(256) k_mark_local(const uint32_t *perm, const uint8_t *maskP, int64_t N, int32_t own0,
                                                    int32_t own1, const int32_t *halo_g, int32_t h0, int32_t h1,
                                                    uint8_t *local)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < N && ((k >= own0 && k < own1) || (maskP[k] & 3))) local[perm[k]] = 1;
    if (k < h1 - h0) local[perm[halo_g[h0 + k]]] = 1;
}

void mark_local(const uint32_t *perm, const uint8_t *maskP, int64_t N, int32_t own0, int32_t own1,
                const int32_t *halo_g, int32_t h0, int32_t h1, uint8_t *local, hipStream_t s)
{
    const int64_t n = N > h1 - h0 ? N : h1 - h0;
    k_mark_local<<<blocks_for(n, 256), 256, 0, s>>>(perm, maskP, N, own0, own1, halo_g, h0, h1, local);
}

// pairs (row node, col node) an element contributes to LOCAL rows: 3 per local corner
__global__ void __launch_bounds__(256) k_csr_pair_count(const int32_t *conn, int64_t E, const uint8_t *local,
                                                        int32_t *cnt)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e > E) return;
    cnt[e] = e < E ? 3 * ((int)local[conn[3 * e]] + (int)local[conn[3 * e + 1]] + (int)local[conn[3 * e + 2]]) : 0;
}

// the same keys and values k_csr_pairs emits, for local rows only, compacted in element order (off = scan of the counts)
__global__ void __launch_bounds__(256) k_csr_pairs_local(const int32_t *conn, int64_t E, const uint8_t *local,
                                                         const int32_t *off, uint64_t *keys, uint32_t *vals)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int32_t n[3] = {conn[3 * e], conn[3 * e + 1], conn[3 * e + 2]};
    int32_t o = off[e];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (!local[n[a]]) continue;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            keys[o] = ((uint64_t)(uint32_t)n[a] << 32) | (uint64_t)(uint32_t)n[b];
            vals[o] = (uint32_t)(9 * e + 3 * a + b);
            ++o;
        }
    }
}

void csr_pair_count(const int32_t *conn, int64_t E, const uint8_t *local, int32_t *cnt, hipStream_t s)
{
    k_csr_pair_count<<<blocks_for(E + 1, 256), 256, 0, s>>>(conn, E, local, cnt);
}

void csr_pairs_local(const int32_t *conn, int64_t E, const uint8_t *local, const int32_t *off, uint64_t *keys,
                     uint32_t *vals, hipStream_t s)
{
    k_csr_pairs_local<<<blocks_for(E, 256), 256, 0, s>>>(conn, E, local, off, keys, vals);
}

__global__ void __launch_bounds__(256) k_csr_heads(const uint64_t *keys, int64_t n, int32_t *head)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    head[k] = (k == 0 || keys[k] != keys[k - 1]) ? 1 : 0;
}

void csr_heads(const uint64_t *keys, int64_t n, int32_t *head, hipStream_t s)
{
    k_csr_heads<<<blocks_for(n, 256), 256, 0, s>>>(keys, n, head);
}

__global__ void __launch_bounds__(256) k_csr_segments(const uint64_t *keys, const int32_t *head, const int32_t *blk,
                                                      int64_t n, int32_t *seg_start, int32_t *brow, int32_t *bcol,
                                                      int32_t *rowcnt)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    if (head[k]) {
        const int32_t b = blk[k];
        const uint64_t key = keys[k];
        seg_start[b] = (int32_t)k;
        brow[b] = (int32_t)(key >> 32);
        bcol[b] = (int32_t)(key & 0xffffffffu);
        atomicAdd(&rowcnt[(int32_t)(key >> 32)], 1);
    }
    if (k == n - 1) seg_start[blk[k] + head[k]] = (int32_t)n;
}

void csr_segments(const uint64_t *keys, const int32_t *head, const int32_t *blk, int64_t n, int32_t *seg_start,
                  int32_t *brow, int32_t *bcol, int32_t *rowcnt, hipStream_t s)
{
    k_csr_segments<<<blocks_for(n, 256), 256, 0, s>>>(keys, head, blk, n, seg_start, brow, bcol, rowcnt);
}

// rows 2i and 2i+1 share node i's block pattern: row 2i holds (2j,2j+1) for each j, then row 2i+1 the same.
__global__ void __launch_bounds__(256) k_csr_export(const int32_t *bptr, const int32_t *bcol, int64_t N,
                                                    int32_t *rowptr, int32_t *col)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int32_t p = bptr[i], cnt = bptr[i + 1] - p;
    rowptr[2 * i] = 4 * p;
    rowptr[2 * i + 1] = 4 * p + 2 * cnt;
    if (i == N - 1) rowptr[2 * N] = 4 * bptr[N];
    for (int k = 0; k < cnt; ++k) {
        const int32_t j = bcol[p + k];
        col[4 * (int64_t)p + 2 * k] = 2 * j;
        col[4 * (int64_t)p + 2 * k + 1] = 2 * j + 1;
        col[4 * (int64_t)p + 2 * cnt + 2 * k] = 2 * j;
        col[4 * (int64_t)p + 2 * cnt + 2 * k + 1] = 2 * j + 1;
    }
}

void csr_export(const int32_t *bptr, const int32_t *bcol, int64_t N, int32_t *rowptr, int32_t *col, hipStream_t s)
{
    k_csr_export<<<blocks_for(N, 256), 256, 0, s>>>(bptr, bcol, N, rowptr, col);
}

// --------------------------------------------- K_ff (solver.rs:365-404) ---
__global__ void __launch_bounds__(256) k_free_flags(const uint8_t *u_known, int64_t n, int32_t *isfree)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i <= n) isfree[i] = (i < n && !u_known[i]) ? 1 : 0; // entry n is the scan's sentinel
}

void free_flags(const uint8_t *u_known, int64_t n, int32_t *isfree, hipStream_t s)
{
    k_free_flags<<<blocks_for(n + 1, 256), 256, 0, s>>>(u_known, n, isfree);
}

__global__ void __launch_bounds__(256) k_reduce_count(const int32_t *bptr, const int32_t *bcol, const double *kval,
                                                      const uint8_t *u_known, int64_t N, int32_t *cnt)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r > 2 * N) return;
    int32_t c = 0;
    if (r < 2 * N && !u_known[r]) {
        const int64_t i = r >> 1;
        const int a = (int)(r & 1);
        const int32_t p = bptr[i], nb = bptr[i + 1] - p;
        const double *row = kval + 4 * (int64_t)p + (int64_t)a * 2 * nb;
        for (int k = 0; k < nb; ++k) {
            const int64_t j = bcol[p + k];
            c += (!u_known[2 * j] && row[2 * k] != 0.0) ? 1 : 0;
            c += (!u_known[2 * j + 1] && row[2 * k + 1] != 0.0) ? 1 : 0;
        }
    }
    cnt[r] = c; // entry 2N is the scan's sentinel (0)
}

void reduce_count(const int32_t *bptr, const int32_t *bcol, const double *kval, const uint8_t *u_known, int64_t N,
                  int32_t *cnt, hipStream_t s)
{
    k_reduce_count<<<blocks_for(2 * N + 1, 256), 256, 0, s>>>(bptr, bcol, kval, u_known, N, cnt);
}

__global__ void __launch_bounds__(256) k_reduce_fill(const int32_t *bptr, const int32_t *bcol, const double *kval,
                                                     const uint8_t *u_known, const int32_t *fidx,
                                                     const int32_t *rowoff, int64_t N, int32_t *rowptr_ff,
                                                     int32_t *col_ff, double *val_ff)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= 2 * N) return;
    if (r == 2 * N - 1) rowptr_ff[fidx[2 * N]] = rowoff[2 * N];
    if (u_known[r]) return;
    const int64_t i = r >> 1;
    const int a = (int)(r & 1);
    const int32_t p = bptr[i], nb = bptr[i + 1] - p;
    const double *row = kval + 4 * (int64_t)p + (int64_t)a * 2 * nb;
    int64_t q = rowoff[r];
    rowptr_ff[fidx[r]] = (int32_t)q;
    for (int k = 0; k < nb; ++k) {
        const int64_t j = bcol[p + k];
        for (int bq = 0; bq < 2; ++bq) {
            const double v = row[2 * k + bq];
            if (!u_known[2 * j + bq] && v != 0.0) {
                col_ff[q] = fidx[2 * j + bq];
                val_ff[q] = v;
                ++q;
            }
        }
    }
}

void reduce_fill(const int32_t *bptr, const int32_t *bcol, const double *kval, const uint8_t *u_known,
                 const int32_t *fidx, const int32_t *rowoff, int64_t N, int32_t *rowptr_ff, int32_t *col_ff,
                 double *val_ff, hipStream_t s)
{
    k_reduce_fill<<<blocks_for(2 * N, 256), 256, 0, s>>>(bptr, bcol, kval, u_known, fidx, rowoff, N, rowptr_ff,
                                                        col_ff, val_ff);
}

} // namespace magk
