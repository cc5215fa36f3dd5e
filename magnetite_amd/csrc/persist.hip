// On-chip conjugate gradients: the whole solve in one launch (see the comment below).  Its own translation unit because
// it is compiled with a different instruction scheduler than the streaming kernels of cg.hip (Makefile).
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>

#include "cg_device.h"
#include "kernels.h"

namespace magk {

// ============================================ on-chip (persistent) CG ===
// When the whole mesh fits the chip -- one workgroup of 512 threads per CU keeps up to four 512-node tiles: r, q and
// the ring words of its nodes in registers; coordinates, the p image (its owned part is the CG vector p), x and the
// halo copies of r, p in LDS -- the CG state never moves through HBM again: ONE launch runs the whole solve.  Per
// iteration a workgroup only publishes q of the nodes other tiles read and its four dot partials, as tagged granules
// (below); every workgroup then sweeps every workgroup's record and the q of its own halo nodes until all tags carry
// the iteration's epoch -- that sweep IS the grid barrier -- and sums the records in one fixed order: the same bits
// in every workgroup, so all of them take the same stop decision in the same iteration.  Spins are bounded (a
// workgroup that gives up sets the timeout word and leaves; the host then falls back to the streaming kernels).
// Same recurrences and state machine as k_cg_fused (alpha, beta from the four exact sums of the previous iterate).
// MG instantiation: several GPUs, each running its tile range, exchanging through per-rank inboxes (further down).
typedef __attribute__((address_space(1))) unsigned int gu32;
// Two shapes of the workgroup, both keeping four 512-node tiles (2048 nodes) per CU:
//   512 threads x 4 nodes per lane: 8 waves = 2 per SIMD, up to 256 VGPRs per lane (round 1-2);
//   768 threads x 3 nodes per lane: 12 waves = 3 per SIMD, 168 VGPRs per lane -- a third wave per SIMD to cover LDS
//       gather latency (round 2's counters: waves parked 0.52 of the time, vector pipe busy 0.49); local node n of the
//       workgroup sits in lane n % 768, slot n / 768, so waves 0-7 carry three nodes per lane and waves 8-11 two, and
//       every SIMD (waves w, w + 4, w + 8) still gets eight node-slots.  Built and measured in round 3 (-DMAG_PERSIST_768,
//       MAG_TUNE_PERSIST_THREADS=768; profiles/r03_persist_phases.json): 11.55 against 11.07 us per iteration at 1M
//       triangles, 6.41 against 5.46 at 100k -- the ring walks are bound by fp64 issue, not by latency (the stamps show
//       the walks of a workgroup taking 7.7 against 7.3 us), and a 12-wave workgroup pays more at every barrier.  Not
//       instantiated in the product.
// Local node n = slot * THREADS + lane belongs to local tile n / B: uniform over a wave (64 | B, 64 | THREADS).
constexpr int kPersistThreadsDefault = 512;

// Inter-workgroup exchange by self-validating granules (CDNA4 guide, Guideline 16 R2: "the data IS the flag"): every
// handed-off 32-bit half travels in its own naturally aligned 8-byte word {value, tag = epoch}; two of them are written
// by one 16-byte write-through store and read by one 16-byte sc1 load, again and again until every tag matches.  No
// arrival counters, no store drains, no fences: a reader can never take a stale or torn value for the current one.  Two buffers by
// parity: nobody can be two epochs ahead of a workgroup that has not finished reading (it would need that workgroup's
// next record first).
typedef __attribute__((address_space(1))) unsigned long long gu64;

// Two 16-byte write-through stores carry the four granules of a value; each 8-byte granule is naturally aligned inside
// its store, so a reader sees every granule whole (and validates each by its tag).
__device__ inline void put_granules(unsigned long long *g, unsigned epoch, double2 v)
{
    unsigned w[4];
    __builtin_memcpy(w, &v, 16);
    const u32x4 a = {w[0], epoch, w[1], epoch}, b = {w[2], epoch, w[3], epoch};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" MAG_WS_DATA
                 :
                 : "v"(g), "v"(a), "v"(b)
                 : "memory");
}

// The same store addressed as SCALAR base + 32-bit byte offset per lane: the hot loop keeps no 64-bit address per node.
__device__ inline void put_granules_at(const unsigned long long *base, uint32_t byte_off, unsigned epoch, double2 v)
{
    unsigned w[4];
    __builtin_memcpy(w, &v, 16);
    const u32x4 a = {w[0], epoch, w[1], epoch}, b = {w[2], epoch, w[3], epoch};
    // MAG_WS_SBASE (s_nop 4) FIRST: the base is an SGPR pair, and when the allocator has it spilled it comes back through v_readlane_b32 --
    // a VALU write of an SGPR, which a VMEM instruction may read only five wait states later.  The compiler's hazard
    // recognizer does not look at the uses inside inline asm, so the asm carries the wait states itself (round 3's
    // diagnostic build had the v_readlane two instructions before the store: a stale base, a wild address, the GPU fault
    // recorded in DESIGN section 4; tests/test_isa_hazards.py scans the emitted ISA for both hazards of these stores).
    asm volatile(MAG_WS_SBASE "global_store_dwordx4 %0, %1, %3 sc1\n\tglobal_store_dwordx4 %0, %2, %3 offset:16 sc1" MAG_WS_DATA
                 :
                 : "v"(byte_off), "v"(a), "v"(b), "s"(base)
                 : "memory");
}

// Four granules (32 bytes) by two 16-byte sc1 loads: each 8-byte granule lies whole inside one store and validates
// itself, so it does not matter that the group is not read atomically.  `base` must be wave-uniform (it becomes the
// buffer resource), the granule group is addressed by the per-lane byte offset.
__device__ inline bool get_granules(const unsigned long long *base, uint32_t bytes, uint32_t off, unsigned epoch,
                                    double2 &v)
{
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, (short)0, (int)bytes, 0x00020000);
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 16);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off + 16, 0, 16);
    const bool ok = a.y == epoch && a.w == epoch && b.y == epoch && b.w == epoch;
    unsigned w[4] = {a.x, a.z, b.x, b.z};
    __builtin_memcpy(&v, w, 16);
    return ok;
}

// Sum over the 64 lanes of a wave by DPP (ALU-rate lane moves; __shfl_down goes through the LDS crossbar, ~10x the
// latency per step, and the on-chip kernel has only two waves per SIMD to hide it).  Inclusive scan inside each row
// of 16 lanes (row_shr 1, 2, 4, 8), then row 0 -> row 1 and row 2 -> row 3 (row_bcast:15), then lane 31 -> rows 2, 3
// (row_bcast:31): the total is in lane 63 and comes back in every lane.  A fixed order, like every sum here.
__device__ inline double wave_sum_dpp(double v)
{
// (BOUND: lanes without a source read 0 by the instruction itself -- with every row enabled the destination then needs no
// zero written into it first, a third of the instructions of a shift step; the broadcast steps enable only some rows and keep
// the explicit 0 for the others)
#define MAG_DPP_STEP(CTRL, ROWMASK, BOUND)                                                                           \
    {                                                                                                                  \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, BOUND);                   \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, BOUND);                   \
        v += __hiloint2double(hi, lo);                                                                                 \
    }
    MAG_DPP_STEP(0x111, 0xf, true)  // row_shr:1
    MAG_DPP_STEP(0x112, 0xf, true)  // row_shr:2
    MAG_DPP_STEP(0x114, 0xf, true)  // row_shr:4
    MAG_DPP_STEP(0x118, 0xf, true)  // row_shr:8
    MAG_DPP_STEP(0x142, 0xa, false) // row_bcast:15 into rows 1 and 3
    MAG_DPP_STEP(0x143, 0xc, false) // row_bcast:31 into rows 2 and 3
#undef MAG_DPP_STEP
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                            __builtin_amdgcn_readlane(__double2loint(v), 63));
}

[[maybe_unused]] constexpr int kStampFrom = 200, kStampTo = 1200;
// 0-6 the phases, 7 sweeps taken; detail (round 4): 8 wave trees of the sums, 9 the sums' barrier (waiting for the workgroup's
// slowest wave), 10 cross-wave chain + record store, 11 the deferred x update, 12 after this wave's sweeps: waiting for the
// workgroup's other waves, 13 record reduction + its barrier
#ifndef MAG_PERSIST_OPAQUE
#define MAG_PERSIST_OPAQUE 1 // the thread index behind an empty asm: 0 nowhere, 1 in the overflow instantiation, 2 everywhere
#endif
#ifndef MAG_PERSIST_OPAQUE_XCHG
#define MAG_PERSIST_OPAQUE_XCHG 1
#endif
#ifndef MAG_PERSIST_OPAQUE_SUM
#define MAG_PERSIST_OPAQUE_SUM 1
#endif
constexpr int kStampPhases = 14;

// a lane's double moved by a DPP control (lanes without a source read 0)
template <int CTRL>
__device__ inline double dpp_move_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Wait for epoch `epoch`: every workgroup's partial record (two 16-byte pieces each, one per thread) and the q of this
// thread's halo nodes, swept together until every tag matches; then the records are summed in one fixed two-level
// order (chunks of eight workgroups, then the chunks) so that all workgroups hold the same bits.  grid <= 256.
// Returns false when the spin budget runs out (some workgroup is not running): the timeout word is set for the host.
template <int NH, bool EB = false, bool OPQ = false>
__device__ inline bool persist_exchange(const PersistParams &P, int par, unsigned epoch, const int32_t (&hg)[NH],
                                        double2 (&hq)[NH], double *s_S, double2 *s_rec, double *s_chunk, double *s_part,
                                        double (&Sx)[4], unsigned long long *stamp = nullptr)
{
    int tid = threadIdx.x;
    if (OPQ && MAG_PERSIST_OPAQUE_XCHG) asm volatile("" : "+v"(tid)); // (recomputed LDS addresses instead of hoisted and spilled ones: see persist_block_sum)
    const int grid = gridDim.x;
    gu32 *tmo = (gu32 *)P.sync + 9;
    // One sweep fetches what is still missing of both: a 16-byte piece of the records per thread and the q of this
    // thread's halo nodes.
    const unsigned long long *recb = P.recg + 8 * (int64_t)par * grid; // this parity's records: 64 bytes per workgroup
    const unsigned long long *qbase = P.qg + 4 * (int64_t)par * P.N;   // ... and q granules: 32 bytes per node
    bool have_rec = tid >= 2 * grid, have_h[NH];
#pragma unroll
    for (int s = 0; s < NH; ++s) {
        have_h[s] = hg[s] < 0;
        hq[s] = make_double2(0.0, 0.0);
    }
    bool done = false;
    // ~0.85 us in all: the other workgroups' records are still on their way, and a sweep that comes too early costs a
    // full round trip.  Two fixed s_sleep instructions, re-tuned in round 2 on one box (us per iteration at 1M triangles):
    // 10+10 10.89, 12+12 10.85, 14+14 10.79, 16+8 10.85, 16+16 10.62, 18+14 10.58, 20+20 10.67, 24+24 10.86; one
    // s_sleep(32) 10.74, one s_sleep(40) 10.79; the same 40 units as a run-time loop of ten s_sleep(4) 10.87.  Round 3, with the
    // shorter walks and half the halo fetched (profiles/r03_persist_ab.txt): 10+6 8.91, 12+10 8.70, 14+8 8.70, 14+10 8.70,
    // 14+12 8.73, 16+10 8.72, 18+14 8.80, 22+16 8.95, 26+20 9.19, one s_sleep(24) 8.72.  With x += alpha p moved into this
    // wait (it takes about half of it): 2 8.34, 4+2 8.20, 6+4 8.23 (4.44 at 100k, where 4+2 gives 4.60), 8+6 8.30, 10+8 8.36.
#ifndef MAG_PERSIST_SLEEP1
#define MAG_PERSIST_SLEEP1 6
#define MAG_PERSIST_SLEEP2 4
#endif
    // The edge-block instantiation reaches this point ~1.5 us earlier in the iteration; re-swept for it (us per iteration at
    // 1M / 100k triangles, one box): 4+2 6.79, 6+4 6.69 / 4.50, 8+6 6.63 / 4.37, 10+8 6.66 / 4.32, 12+10 6.73 / 4.39, 14+12 6.81.
    // After the workgroup's final sums became a cross-lane chain (persist_block_sum: the record leaves ~0.4 us earlier):
    // 2+0 6.40 / 4.28, 4+2 6.11 / 4.26, 4+4 6.09 / 4.12, 6+4 6.09 / 4.05, 8+6 6.15 / 4.09, 10+8 6.23 / 4.14, 12+10 6.32 / 4.26.
#ifndef MAG_PERSIST_SLEEP1_EB
#define MAG_PERSIST_SLEEP1_EB 6
#define MAG_PERSIST_SLEEP2_EB 4
#endif
    __builtin_amdgcn_s_sleep(EB ? MAG_PERSIST_SLEEP1_EB : MAG_PERSIST_SLEEP1);
    if ((EB ? MAG_PERSIST_SLEEP2_EB : MAG_PERSIST_SLEEP2) > 0) __builtin_amdgcn_s_sleep(EB ? MAG_PERSIST_SLEEP2_EB : MAG_PERSIST_SLEEP2);
#ifdef MAG_PERSIST_STAMPS
    if (stamp) stamp[0] = __builtin_amdgcn_s_memrealtime();
#endif
#ifndef MAG_PERSIST_POLL
#define MAG_PERSIST_POLL 1
#endif
#if MAG_PERSIST_POLL
    // Every WAVE polls for what its own lanes still miss and the workgroup meets once, when every wave has everything:
    // a workgroup barrier per sweep (round 1-2) made every sweep as slow as the slowest wave's loads and started the next
    // one only after all of them.
    bool wave_ok = false;
    unsigned spins = 0;
    for (; spins < P.spin_limit; ++spins) {
        bool ok = true;
        if (!have_rec) {
            double2 v;
            have_rec = get_granules(recb, 64u * (uint32_t)grid, 32u * (uint32_t)tid, epoch, v);
            if (have_rec) s_rec[tid] = v;
            ok = have_rec;
        }
#pragma unroll
        for (int s = 0; s < NH; ++s)
            if (!have_h[s]) {
                have_h[s] = get_granules(qbase, 32u * (uint32_t)P.N, 32u * (uint32_t)hg[s], epoch, hq[s]);
                ok &= have_h[s];
            }
        if (__all(ok ? 1 : 0)) {
            wave_ok = true;
            break;
        }
        if ((spins & 255u) == 255u) { // somebody gave up: do not wait for a grid that will never be complete
            const unsigned dead = (tid & 63) == 0 ? __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (__any(dead != 0u ? 1 : 0)) break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
#ifdef MAG_PERSIST_STAMPS
    if (stamp) stamp[4] = __builtin_amdgcn_s_memrealtime(); // this wave has its pieces
#endif
    // every wave has everything?  One flag word per wave, ONE barrier (also: s_rec complete), one 32-byte read: the
    // library's __syncthreads_and is three barriers around an LDS atomic.
    {
        constexpr int NWV = 8; // (512 threads; the 768-thread shape takes the library call)
        if (blockDim.x == 64 * NWV) {
            uint32_t *s_flag = (uint32_t *)(s_chunk + 4); // words 4 .. 7 of s_chunk: free (0, 1: best_param; 16 ..: stamps)
            if ((tid & 63) == 0) s_flag[tid >> 6] = wave_ok ? 1u : 0u;
            __syncthreads();
            const uint4 f0 = ((const uint4 *)s_flag)[0], f1 = ((const uint4 *)s_flag)[1];
            done = (f0.x & f0.y & f0.z & f0.w & f1.x & f1.y & f1.z & f1.w) != 0u;
#ifdef MAG_PERSIST_STAMPS
            if (stamp) stamp[3] = __builtin_amdgcn_s_memrealtime(); // every wave of the workgroup has its pieces
#endif
        } else {
            done = __syncthreads_and(wave_ok ? 1 : 0) != 0;
        }
    }
#ifdef MAG_PERSIST_STAMPS
    if (stamp) {
        stamp[1] = __builtin_amdgcn_s_memrealtime();
        stamp[2] = spins + 1;
    }
#endif
#else
    for (unsigned spins = 0; spins < P.spin_limit; ++spins) {
        bool ok = true;
        if (!have_rec) {
            double2 v;
            have_rec = get_granules(recb, 64u * (uint32_t)grid, 32u * (uint32_t)tid, epoch, v);
            if (have_rec) s_rec[tid] = v;
            ok = have_rec;
        }
#pragma unroll
        for (int s = 0; s < NH; ++s)
            if (!have_h[s]) {
                have_h[s] = get_granules(qbase, 32u * (uint32_t)P.N, 32u * (uint32_t)hg[s], epoch, hq[s]);
                ok &= have_h[s];
            }
        if (__syncthreads_and(ok ? 1 : 0)) {
            done = true;
#ifdef MAG_PERSIST_STAMPS
            if (stamp) {
                stamp[1] = __builtin_amdgcn_s_memrealtime();
                stamp[2] = spins + 1;
            }
#endif
            break;
        }
        if ((spins & 255u) == 255u) { // somebody else gave up: do not wait for a grid that will never be complete
            const int dead =
                tid == 0 && __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ? 1 : 0;
            if (__syncthreads_or(dead)) break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads(); // s_rec complete
#endif
    if (!done) {
        if (tid == 0) __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    // every workgroup sums the same values in the same order: same bits everywhere.
    // Round 4 (512 threads): one tree for the four sums, as in persist_block_sum.  Wave w takes records 32 w .. 32 w + 31, two
    // lanes per record: an even lane loads the piece {sum 0, sum 1}, an odd lane {sum 2, sum 3} -- where level 1 of that tree
    // leaves its values -- then level 2, two row shifts, the rows' partials through LDS, and EVERY wave adds the 32 partials of
    // each sum itself (one 16-byte read, a four-step tree per row): the sums arrive in scalar registers of every wave, with
    // one barrier and no broadcast through s_S (round 3: four loads per lane, a six-level tree, s_S, a second barrier).
    if (blockDim.x == 512) {
        const int l = tid & 63, m = 32 * (tid >> 6) + (l >> 1);
        const double2 pc = m < grid ? s_rec[2 * m + (l & 1)] : make_double2(0.0, 0.0);
        const bool two = (l & 2) != 0;
        const double k2 = two ? pc.y : pc.x, s2 = two ? pc.x : pc.y;
        double v = k2 + dpp_move_f64<0x4E>(s2);
        v += dpp_move_f64<0x114>(v);
        v += dpp_move_f64<0x118>(v);
        if ((l & 15) >= 12) s_part[(((l & 1) << 1) | ((l >> 1) & 1)) * 32 + (tid >> 6) * 4 + (l >> 4)] = v;
        __syncthreads();
        const double2 pr = ((const double2 *)(s_part + (l >> 4) * 32))[l & 15];
        double t = pr.x + pr.y;
        t += dpp_move_f64<0x111>(t);
        t += dpp_move_f64<0x112>(t);
        t += dpp_move_f64<0x114>(t);
        t += dpp_move_f64<0x118>(t);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            Sx[c] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t), 16 * c + 15),
                                     __builtin_amdgcn_readlane(__double2loint(t), 16 * c + 15));
        return true;
    }
    if (tid < 256) {
        const int c = tid >> 6, lane = tid & 63;
        const double *rec = (const double *)s_rec; // record m: doubles 4 m .. 4 m + 3
        // (grid <= 256: at most four records per lane -- their loads in flight together, the additions in the loop's order)
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = lane + 64 * k < grid ? rec[4 * (lane + 64 * k) + c] : 0.0;
        double S = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) S = lane + 64 * k < grid ? S + v[k] : S;
        S = wave_sum_dpp(S);
        if (lane == 0) s_S[c] = S;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) Sx[c] = s_S[c];
    return true;
}

// ---- multi-GPU: the same granules through a window of HOST memory every rank has mapped (system scope) ----
__device__ inline void put_granules_sys(unsigned long long *g, unsigned tag, double2 v)
{
    unsigned w[4];
    __builtin_memcpy(w, &v, 16);
    const u32x4 a = {w[0], tag, w[1], tag}, b = {w[2], tag, w[3], tag};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc0 sc1" MAG_WS_DATA
                 :
                 : "v"(g), "v"(a), "v"(b)
                 : "memory");
}

// `base` wave-uniform, as for get_granules; sc0 | sc1 = system scope: the loads go to the host memory itself.
__device__ inline bool get_granules_sys(const unsigned long long *base, uint32_t bytes, uint32_t off, unsigned tag,
                                        double2 &v)
{
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, (short)0, (int)bytes, 0x00020000);
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 17);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off + 16, 0, 17);
    const bool ok = a.y == tag && a.w == tag && b.y == tag && b.w == tag;
    unsigned w[4] = {a.x, a.z, b.x, b.z};
    __builtin_memcpy(&v, w, 16);
    return ok;
}

// q of an interface node this rank owns, into the inbox of every rank that reads it
// system-scope granules at SCALAR base + 32-bit byte offset (no 64-bit address per node kept across the loop)
__device__ inline void put_granules_sys_at(const uint8_t *base, uint32_t byte_off, unsigned tag, double2 v)
{
    unsigned w[4];
    __builtin_memcpy(w, &v, 16);
    const u32x4 a = {w[0], tag, w[1], tag}, b = {w[2], tag, w[3], tag};
    // (s_nop 4 first: see put_granules_at -- an SGPR base restored by v_readlane needs five wait states before a VMEM read)
    asm volatile(MAG_WS_SBASE "global_store_dwordx4 %0, %1, %3 sc0 sc1\n\tglobal_store_dwordx4 %0, %2, %3 offset:16 sc0 sc1" MAG_WS_DATA
                 :
                 : "v"(byte_off), "v"(a), "v"(b), "s"(base)
                 : "memory");
}

// `readers`: P.iface_readers[slot], loaded ONCE per solve by the caller.  Loaded here, the loop below would start with a wait
// for the memory counter at the top of every round -- and on gfx9 that counter holds the stores as well: every rank's
// store would wait for the previous one's round trip across xGMI.
__device__ inline void publish_q(const PersistParams &P, int par, int32_t slot, uint32_t readers, unsigned tag, double2 v)
{
    // (an inbox is 64 + 128 R + 64 n_iface bytes: far below 4 GB)
    const uint32_t off = 64u + 128u * (uint32_t)P.nranks + 32u * ((uint32_t)par * (uint32_t)P.n_iface + (uint32_t)slot);
    if (P.win_shared) {
        put_granules_sys_at(P.inbox[0], off, tag, v);
        return;
    }
    for (int r = 0; r < P.nranks; ++r)
        if ((readers >> r) & 1u) put_granules_sys_at(P.inbox[r], off, tag, v);
}

// Exchange of the multi-GPU kernel.  Halo q: from this GPU's granules, or from the window when another rank owns the
// node (hg < 0 encodes the interface slot as -2 - slot).  Sums: workgroup 0 gathers this rank's records, publishes
// their sum in the window, gathers every rank's record there, sums them in rank order and republishes the result on
// the device; all other workgroups wait for that republished record.  Every workgroup on every rank ends with the
// same bits.
template <int NH>
__device__ inline bool persist_exchange_mg(const PersistParams &P, int par, unsigned tag, const int32_t (&hg)[NH],
                                           double2 (&hq)[NH], double *s_S, double2 *s_rec,
                                           [[maybe_unused]] unsigned long long *stamp = nullptr)
{
    const int tid = threadIdx.x;
    const int grid = (int)gridDim.x - P.comm_wg, R = P.nranks; // compute workgroups
    gu32 *tmo = (gu32 *)P.sync + 9;
    uint8_t *mine = P.inbox[P.rank]; // this rank's inbox: the only one it reads
    gu32 *wtmo = (gu32 *)mine;
    auto inbox_rec = [&](int r) { return (unsigned long long *)(P.inbox[r] + 64); };
    const unsigned long long *recb = P.recg + 8 * (int64_t)par * grid;
    const unsigned long long *qbase = P.qg + 4 * (int64_t)par * P.N;
    const unsigned long long *wq =
        (const unsigned long long *)(mine + 64 + 128 * (size_t)R) + 4 * (int64_t)par * P.n_iface;
    const bool lead = blockIdx.x == 0 && !P.comm_wg; // with an exchange workgroup nobody leads: all wait for its record
    bool have_h[NH];
#pragma unroll
    for (int e = 0; e < NH; ++e) {
        have_h[e] = hg[e] == -1;
        hq[e] = make_double2(0.0, 0.0);
    }
    auto gave_up = [&](unsigned spins) { // uniform: every 256th spin one lane looks at the two timeout words
        if ((spins & 255u) != 255u) return false;
        const int dead = tid == 0 && (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
                                      __hip_atomic_load(wtmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)
                             ? 1 : 0;
        return __syncthreads_or(dead) != 0;
    };
    auto fail = [&]() {
        if (tid == 0) __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < (P.win_shared ? 1 : R)) // tell every rank
            __hip_atomic_store((gu32 *)P.inbox[tid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return false;
    };
    // Window reads cross PCIe: they are issued only once the grid-wide sums are known -- every rank wrote its q before
    // its record (posted writes of one device stay in order), so by then they are there and ONE read per value does
    // (each granule still validates itself; a miss is simply read again).
    auto fetch_halo = [&](bool &ok, bool window) {
#pragma unroll
        for (int e = 0; e < NH; ++e)
            if (!have_h[e]) {
                if (hg[e] >= 0)
                    have_h[e] = get_granules(qbase, 32u * (uint32_t)P.N, 32u * (uint32_t)hg[e], tag, hq[e]);
                else if (window)
                    have_h[e] = get_granules_sys(wq, 32u * (uint32_t)P.n_iface, 32u * (uint32_t)(-2 - hg[e]), tag, hq[e]);
                ok &= have_h[e] || (hg[e] < 0 && !window);
            }
    };
    auto fetch_window_halo = [&]() { // after the sums: bounded retries
        for (unsigned spins = 0; spins < P.spin_limit; ++spins) {
            bool ok = true;
            fetch_halo(ok, true);
            if (__syncthreads_and(ok ? 1 : 0)) return true;
            if (gave_up(spins)) break;
            __builtin_amdgcn_s_sleep(2);
        }
        return false;
    };
#ifndef MAG_PERSIST_SLEEP_MG
#define MAG_PERSIST_SLEEP_MG 20
#endif
    __builtin_amdgcn_s_sleep(MAG_PERSIST_SLEEP_MG);
    __builtin_amdgcn_s_sleep(MAG_PERSIST_SLEEP_MG);
#ifdef MAG_PERSIST_STAMPS
    if (stamp) stamp[0] = __builtin_amdgcn_s_memrealtime();
#endif
    if (lead) {
        // (1) this rank's records (and this workgroup's own halo values)
        bool have_rec = tid >= 2 * grid, done = false;
        unsigned spins = 0;
        for (; spins < P.spin_limit; ++spins) {
            bool ok = true;
            if (!have_rec) {
                double2 v;
                have_rec = get_granules(recb, 64u * (uint32_t)grid, 32u * (uint32_t)tid, tag, v);
                if (have_rec) s_rec[tid] = v;
                ok = have_rec;
            }
            fetch_halo(ok, false);
            if (__syncthreads_and(ok ? 1 : 0)) {
                done = true;
                break;
            }
            if (gave_up(spins)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!done) return fail();
        if (tid < 64) {
            double S[4] = {0.0, 0.0, 0.0, 0.0};
            for (int m = tid; m < grid; m += 64) {
                const double2 a = s_rec[2 * m], b = s_rec[2 * m + 1];
                S[0] += a.x;
                S[1] += a.y;
                S[2] += b.x;
                S[3] += b.y;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) S[c] = wave_sum_dpp(S[c]);
            // (2) this rank's sums into every rank's inbox (two pieces each; one store serves all in a shared window)
            if (tid < 2 * (P.win_shared ? 1 : R))
                put_granules_sys(inbox_rec(tid >> 1) + 4 * (2 * ((int64_t)par * R + P.rank) + (tid & 1)), tag,
                                 (tid & 1) == 0 ? make_double2(S[0], S[1]) : make_double2(S[2], S[3]));
        }
        __syncthreads();
        // (3) every rank's sums from the window
        bool have_w = tid >= 2 * R;
        done = false;
        for (; spins < P.spin_limit; ++spins) {
            if (!have_w) {
                double2 v;
                have_w = get_granules_sys(inbox_rec(P.rank) + 8 * (int64_t)par * R, 64u * (uint32_t)R, 32u * (uint32_t)tid,
                                          tag, v);
                if (have_w) s_rec[tid] = v;
            }
            if (__syncthreads_and(have_w ? 1 : 0)) {
                done = true;
                break;
            }
            if (gave_up(spins)) break;
            __builtin_amdgcn_s_sleep(4);
        }
        if (!done) return fail();
        if (tid < 4) { // rank order: the same bits on every rank
            const double *rec = (const double *)s_rec;
            double t = 0.0;
            for (int r = 0; r < R; ++r) t += rec[4 * r + tid];
            s_S[tid] = t;
        }
        __syncthreads();
        // (4) republished for the other workgroups of this GPU
        if (tid < 2)
            put_granules(P.grec + 4 * (2 * par + tid), tag,
                         tid == 0 ? make_double2(s_S[0], s_S[1]) : make_double2(s_S[2], s_S[3]));
        return fetch_window_halo() ? true : fail();
    }
    // exchange workgroup present: the rank sums land in every inbox, this rank's own included, and every workgroup reads
    // them there itself (one hop fewer than waiting for a republished total: 9.1 against 9.8 us per iteration, eight
    // ranks on one GPU); otherwise workgroup 0 republishes the total on the device
    const bool direct = P.comm_wg != 0;
    bool have_g = tid >= (direct ? 2 * R : 2), done = false;
    [[maybe_unused]] unsigned sweeps_ = 0;
    for (unsigned spins = 0; spins < P.spin_limit; ++spins) {
        ++sweeps_;
        bool ok = true;
        if (!have_g) {
            double2 v;
            have_g = direct ? get_granules_sys(inbox_rec(P.rank) + 8 * (int64_t)par * R, 64u * (uint32_t)R,
                                               32u * (uint32_t)tid, tag, v)
                            : get_granules(P.grec, 64u * 2u, 64u * (uint32_t)par + 32u * (uint32_t)tid, tag, v);
            if (have_g) s_rec[tid] = v;
            ok = have_g;
        }
        // device inboxes: the interface q of this workgroup's halo sit in LOCAL memory and were stored before their
        // owners' records were even summed -- they are fetched in the same sweeps as the republished sums, not after them
        // (a host window is read only once the sums are there: every miss would be a PCIe round trip)
        fetch_halo(ok, !P.win_shared);
        if (__syncthreads_and(ok ? 1 : 0)) {
            done = true;
            break;
        }
        if (gave_up(spins)) break;
        __builtin_amdgcn_s_sleep(2);
    }
    if (!done) return fail();
#ifdef MAG_PERSIST_STAMPS
    if (stamp) { // (a workgroup barrier per sweep here: no wave finishes before the others)
        stamp[1] = stamp[3] = stamp[4] = __builtin_amdgcn_s_memrealtime();
        stamp[2] = sweeps_;
    }
#endif
    if (tid < 4) {
        const double *rec = (const double *)s_rec;
        double t = rec[tid];
        if (direct) { // rank order, as the exchange workgroup adds them: the same bits
            t = 0.0;
            for (int r = 0; r < R; ++r) t += rec[4 * r + tid];
        }
        s_S[tid] = t;
    }
    __syncthreads();
    if (!P.win_shared) return true;
    return fetch_window_halo() ? true : fail();
}

// The EXCHANGE WORKGROUP of the multi-GPU kernel (device inboxes only; PersistParams::comm_wg): one extra workgroup that
// holds no tile and does nothing but the rank-level exchange workgroup 0 would otherwise carry on top of its four
// tiles.  It has nothing else to do, so it polls without the initial wait: per epoch it (1) gathers this rank's partial
// records as they land, (2) sums them in the fixed order and stores the rank's sum into every rank's inbox, (3) gathers
// every rank's sum from its own inbox and (4) adds them in rank order, as every compute workgroup of this GPU does with the
// same R sums -- so it takes the same stop decision from the same bits and leaves when they do.
__device__ inline void persist_comm_loop(const PersistParams &P, double *s_S, double2 *s_rec)
{
    const int tid = threadIdx.x;
    const int grid = (int)gridDim.x - 1, R = P.nranks;
    gu32 *tmo = (gu32 *)P.sync + 9;
    uint8_t *mine = P.inbox[P.rank];
    gu32 *wtmo = (gu32 *)mine;
    auto inbox_rec = [&](int r) { return (unsigned long long *)(P.inbox[r] + 64); };
    auto gave_up = [&](unsigned spins) {
        if ((spins & 255u) != 255u) return false;
        const int dead = tid == 0 && (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
                                      __hip_atomic_load(wtmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)
                             ? 1 : 0;
        return __syncthreads_or(dead) != 0;
    };
    auto fail = [&]() {
        if (tid == 0) __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < R) __hip_atomic_store((gu32 *)P.inbox[tid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    int par = 0;
    unsigned tag = P.tag_base + 1u;
    double target = P.tol, bb = 0.0;
#ifdef MAG_PERSIST_STAMPS
    // phases of the exchange workgroup, iterations [kStampFrom, kStampTo): waiting for this rank's records / summing them and
    // storing the rank's sum into every inbox / waiting for every rank's sum / the rest (its row of PersistParams::stamps is
    // the one behind the compute workgroups': words 0-3, word kStampPhases the iterations counted)
    unsigned long long cs_[4] = {0, 0, 0, 0}, cn_ = 0;
#endif
    for (long long j = 0;; ++j) {
#ifdef MAG_PERSIST_STAMPS
        const bool cst_ = tid == 0 && j >= kStampFrom && j < kStampTo;
        unsigned long long ct0_ = 0, ct1_ = 0, ct2_ = 0, ct3_ = 0;
        if (cst_) ct0_ = __builtin_amdgcn_s_memrealtime();
#endif
        const unsigned long long *recb = P.recg + 8 * (int64_t)par * grid;
        // (1) this rank's records
        bool have_rec = tid >= 2 * grid, done = false;
        unsigned spins = 0;
        for (; spins < P.spin_limit; ++spins) {
            if (!have_rec) {
                double2 v;
                have_rec = get_granules(recb, 64u * (uint32_t)grid, 32u * (uint32_t)tid, tag, v);
                if (have_rec) s_rec[tid] = v;
            }
            if (__syncthreads_and(have_rec ? 1 : 0)) {
                done = true;
                break;
            }
            if (gave_up(spins)) break;
        }
        if (!done) return fail();
#ifdef MAG_PERSIST_STAMPS
        if (cst_) ct1_ = __builtin_amdgcn_s_memrealtime();
#endif
        if (tid < 64) {
            double S[4] = {0.0, 0.0, 0.0, 0.0};
            for (int m = tid; m < grid; m += 64) {
                const double2 a = s_rec[2 * m], b = s_rec[2 * m + 1];
                S[0] += a.x;
                S[1] += a.y;
                S[2] += b.x;
                S[3] += b.y;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) S[c] = wave_sum_dpp(S[c]);
            // (2) into every rank's inbox, this rank's own included
            if (tid < 2 * R)
                put_granules_sys(inbox_rec(tid >> 1) + 4 * (2 * ((int64_t)par * R + P.rank) + (tid & 1)), tag,
                                 (tid & 1) == 0 ? make_double2(S[0], S[1]) : make_double2(S[2], S[3]));
        }
        __syncthreads();
#ifdef MAG_PERSIST_STAMPS
        if (cst_) ct2_ = __builtin_amdgcn_s_memrealtime();
#endif
        // (3) every rank's sum
        bool have_w = tid >= 2 * R;
        done = false;
        for (; spins < P.spin_limit; ++spins) {
            if (!have_w) {
                double2 v;
                have_w = get_granules_sys(inbox_rec(P.rank) + 8 * (int64_t)par * R, 64u * (uint32_t)R, 32u * (uint32_t)tid,
                                          tag, v);
                if (have_w) s_rec[tid] = v;
            }
            if (__syncthreads_and(have_w ? 1 : 0)) {
                done = true;
                break;
            }
            if (gave_up(spins)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!done) return fail();
#ifdef MAG_PERSIST_STAMPS
        if (cst_) {
            ct3_ = __builtin_amdgcn_s_memrealtime();
            cs_[0] += ct1_ - ct0_;
            cs_[1] += ct2_ - ct1_;
            cs_[2] += ct3_ - ct2_;
            ++cn_;
        }
#endif
        if (tid < 4) { // (4) rank order: the same bits on every rank
            const double *rec = (const double *)s_rec;
            double t = 0.0;
            for (int r = 0; r < R; ++r) t += rec[4 * r + tid];
            s_S[tid] = t;
        }
        __syncthreads();
        // the compute workgroups read the same R sums in this inbox and add them in the same order; their stop decision,
        // from the same bits (k_cg_persist, top of its loop), is taken here too
        const double rr = s_S[0];
        if (j == 0) {
            bb = rr;
            target = P.stop_mode == 2 ? P.tol * sqrt(bb) : P.tol;
        }
        const double cost = P.stop_mode == 1 ? fabs(rr) : sqrt(rr);
        const long long it_done = j - 1;
        if ((j == 0 && bb == 0.0) || (it_done >= 1 && cost <= target) || !(fabs(rr) <= 1.79769313486231570e308) ||
            it_done >= P.max_iter) {
#ifdef MAG_PERSIST_STAMPS
            if (tid == 0 && P.stamps) {
                unsigned long long *o = P.stamps + (size_t)grid * (kStampPhases + 1);
                for (int k = 0; k < 3; ++k) o[k] = cs_[k];
                o[kStampPhases] = cn_;
            }
#endif
            return;
        }
        __syncthreads(); // s_S and s_rec are rewritten by the next epoch
        par ^= 1;
        ++tag;
    }
}

// Workgroup totals of four partial sums for the two publishing threads (0 and 1).
// Round 4 (512 threads): the in-kernel stamps put 0.37 us into four full DPP wave trees and 0.49 us into the seven-step
// cross-wave chain + the record's store -- a microsecond between the last walk and the record leaving, every iteration.
// Now ONE tree serves the four sums: level 1 (lane pairs) halves the values a lane carries from four to two -- even lanes
// keep sums 0 and 1, odd lanes 2 and 3 --, level 2 (pairs of pairs) from two to one, two row shifts finish a row of 16 lanes;
// the four rows' partials of every wave go to LDS (s_part[sum][wave * 4 + row]: 32 per sum) and after the barrier wave 0
// adds them with one 16-byte read and a four-step tree per row -- 7 + 5 additions on the critical path where there were
// 24 + 7, a fixed order of additions as before (so every run gives the same bits; they are not round 3's bits).
constexpr int kPersistPartDoubles = 128; // s_part: four sums x (8 waves x 4 rows)
template <int THREADS, bool OPQ = false>
__device__ inline void persist_block_sum(double (&acc)[4], double *s_red, double *s_part,
                                         [[maybe_unused]] unsigned long long *sub = nullptr)
{
    constexpr int kPersistThreads = THREADS;
    constexpr int NWV = kPersistThreads / 64;
    if (NWV == 8) {
        // (tx: the thread index behind an empty asm -- the LDS addresses below are then recomputed in every iteration, three
        // instructions each.  Taken from threadIdx.x directly they are loop-invariant: the compiler hoisted them out of the CG
        // loop, had no registers to keep them in -- the edge-block kernel sits at 256 of 256 -- and spilled them: nine
        // scratch reloads per iteration, each behind an s_waitcnt vmcnt(0), i.e. behind the granule stores in flight.)
        int tx = threadIdx.x;
        if (OPQ && MAG_PERSIST_OPAQUE_SUM) asm volatile("" : "+v"(tx));
        const int l = tx & 63;
        const bool odd = (l & 1) != 0, two = (l & 2) != 0;
        // level 1, lanes l and l ^ 1 (quad_perm [1, 0, 3, 2]): an even lane keeps sums 0, 1 and hands over 2, 3; an odd lane the reverse
        const double ka = odd ? acc[2] : acc[0], kb = odd ? acc[3] : acc[1];
        const double sa = odd ? acc[0] : acc[2], sb = odd ? acc[1] : acc[3];
        const double a = ka + dpp_move_f64<0xB1>(sa), b = kb + dpp_move_f64<0xB1>(sb);
        // level 2, lanes l and l ^ 2 (quad_perm [2, 3, 0, 1]): lanes 0, 1 of a quad keep a, lanes 2, 3 keep b
        const double k2 = two ? b : a, s2 = two ? a : b;
        double v = k2 + dpp_move_f64<0x4E>(s2); // lane l of a quad now holds the quad's total of sum {0, 2, 1, 3}[l & 3]
        v += dpp_move_f64<0x114>(v);            // row_shr:4
        v += dpp_move_f64<0x118>(v);            // row_shr:8: lanes 12-15 of every row hold the row's totals
        if ((l & 15) >= 12) s_part[(((l & 1) << 1) | ((l >> 1) & 1)) * 32 + (tx >> 6) * 4 + (l >> 4)] = v;
#ifdef MAG_PERSIST_STAMPS
        if (sub) sub[0] = __builtin_amdgcn_s_memrealtime(); // wave trees done, at the barrier
#endif
        __syncthreads();
#ifdef MAG_PERSIST_STAMPS
        if (sub) sub[1] = __builtin_amdgcn_s_memrealtime(); // every wave has arrived
#endif
        if (tx < 64) { // row c of wave 0 adds sum c's 32 partials: two per lane, then a tree over the row
            const double2 pr = ((const double2 *)(s_part + (l >> 4) * 32))[l & 15];
            double t = pr.x + pr.y;
            t += dpp_move_f64<0x111>(t);
            t += dpp_move_f64<0x112>(t);
            t += dpp_move_f64<0x114>(t);
            t += dpp_move_f64<0x118>(t);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[c] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t), 16 * c + 15),
                                          __builtin_amdgcn_readlane(__double2loint(t), 16 * c + 15));
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = wave_sum_dpp(acc[c]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) s_red[c * NWV + (threadIdx.x >> 6)] = acc[c];
    }
    __syncthreads();
    if (threadIdx.x < 2) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < NWV; ++i) t += s_red[c * NWV + i];
            acc[c] = t;
        }
    }
}

constexpr int kPersistRegs = 5; // ring words in registers per node: 10 entries, a closed fan of valence <= 9
#ifndef MAG_PERSIST_BLOCK
#define MAG_PERSIST_BLOCK 1 // (uncached walks only, i.e. the multi-GPU instantiation) measured in round 3, us per iteration at 1M
#endif                      // triangles: no block 11.06-11.10, 5: 11.02, 4+3: 11.19-11.21, 5+2: 11.14; one block of 7 only compiles
                            // without iterative-ilp: 11.41-11.71 (profiles/r03_persist_ab.txt).  Back to 1: with the
                            // tile-relative slots below ROCm 7.2's iterative-ilp scheduler crashes on any block.
#ifndef MAG_PERSIST_BLOCK2
#define MAG_PERSIST_BLOCK2 MAG_PERSIST_BLOCK
#endif
// ring entries walked as straight-line blocks: 1 .. BLOCK - 1 and BLOCK .. BLOCK2 - 1 (cg_device.h, ring_walk_uniform)
constexpr int kPersistBlock = MAG_PERSIST_BLOCK, kPersistBlock2 = MAG_PERSIST_BLOCK2;
#ifndef MAG_PERSIST_WEIGHTS
#define MAG_PERSIST_WEIGHTS 6
#endif
#ifndef MAG_PERSIST_SIBLINGS
#define MAG_PERSIST_SIBLINGS 1
#endif
constexpr bool kPersistSiblings = MAG_PERSIST_SIBLINGS != 0; // sibling tiles of a workgroup read each other's LDS slots
#ifndef MAG_PERSIST_DEFER_X
#define MAG_PERSIST_DEFER_X 1
#endif
constexpr bool kPersistDeferX = MAG_PERSIST_DEFER_X != 0; // x += alpha p in the exchange's first wait (see the loop)
#ifndef MAG_PERSIST_PRIO
#define MAG_PERSIST_PRIO 3 // waves 4-7 (the arbitration losers of their SIMDs) take priority for their last two node slots: 9.55 -> 9.21 us
#endif
// triangle weights c0 / (2A) kept in registers per node (cg_device.h, ring_walk_cached); 0: recomputed every iteration.
// The multi-GPU instantiation has no registers to spare for them (237 VGPRs without).
constexpr int kPersistWeights = MAG_PERSIST_WEIGHTS;
#ifndef MAG_PERSIST_EDGE_BLOCKS
#define MAG_PERSIST_EDGE_BLOCKS 1
#endif
constexpr bool kPersistEdgeBlocks = MAG_PERSIST_EDGE_BLOCKS != 0; // edge blocks in registers instead of triangle weights
#ifndef MAG_PERSIST_NB
#define MAG_PERSIST_NB 6
#endif
#ifndef MAG_PERSIST_SADDR
#define MAG_PERSIST_SADDR 1
#endif
#ifndef MAG_PERSIST_ENT_SCALAR
#define MAG_PERSIST_ENT_SCALAR 1
#endif
#ifndef MAG_PERSIST_PACK_MG
#define MAG_PERSIST_PACK_MG 1
#endif
#ifndef MAG_PERSIST_EB_PACK
#define MAG_PERSIST_EB_PACK 1
#endif
constexpr int kPersistBlockEntries = MAG_PERSIST_NB; // block entries per node: a closed fan of valence 6 is exactly six blocks
constexpr int kPersistNh = 2;   // halo entries per thread: a workgroup's tiles may carry 2 * THREADS halo nodes in all
#ifndef MAG_PERSIST_ONE_TILE
#define MAG_PERSIST_ONE_TILE 1
#endif
constexpr bool kPersistOneTile = MAG_PERSIST_ONE_TILE != 0;
constexpr int persist_npt(int threads) { return threads == 768 ? 3 : 4; } // nodes per lane

// Phase stamps (diagnostic build only: -DMAG_PERSIST_STAMPS, scripts/persist_phases.sh): lane 0 of every workgroup reads
// the 100 MHz constant clock at the phase boundaries of iterations [kStampFrom, kStampTo) and adds the differences up
// in registers; at the end it writes them to PersistParams::stamps (memory nothing else in the kernel reads; no output
// value is computed from them).  In the product build no stamp executes and the buffer is never touched.
#ifdef MAG_PERSIST_STAMPS
#define MAG_STAMP(k)                                                                                                   \
    if (stamping) {                                                                                                    \
        const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();                                              \
        stamp_sum[k] += now_ - stamp_last;                                                                             \
        stamp_last = now_;                                                                                             \
    }
#else
#define MAG_STAMP(k)
#endif
#ifndef MAG_PERSIST_STAMP_TID
#define MAG_PERSIST_STAMP_TID 0 // the lane that stamps (0: wave 0, the older wave of its SIMD; 256: wave 4, its partner)
#endif


// the per-slot flag bytes of a lane's nodes in ONE register (the on-chip kernel has none to spare)
template <int N>
struct PackedFlags {
    static_assert(N <= 4, "one byte per node slot");
    uint32_t v = 0;
    __device__ uint32_t operator[](int s) const { return (v >> (8 * s)) & 0xffu; }
    __device__ void set(int s, uint32_t f) { v = (v & ~(0xffu << (8 * s))) | ((f & 0xffu) << (8 * s)); }
};

// a value every lane holds identically, moved to scalar registers
__device__ inline double uniform_f64(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                            __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// ONE workgroup holds the whole mesh (up to four tiles: the size of the reference's own examples): its own four sums are the
// grid's, nothing is read through memory (no tile of another workgroup exists: no halo entry is live), and the exchange -- a
// store-to-load round trip through the memory side, ~1.7 us of an iteration's ~3.9 at this size -- shrinks to one barrier.
// s_S: the workgroup's sums, left there by thread 0 where the record is published (they are not kept in registers up to here: the
// edge-block kernel has none to spare).
template <int NH>
__device__ inline void persist_single_workgroup(double *s_S, double (&Sx)[4], double2 (&hq)[NH])
{
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) Sx[c] = uniform_f64(s_S[c]);
#pragma unroll
    for (int e = 0; e < NH; ++e) hq[e] = make_double2(0.0, 0.0);
    __syncthreads(); // (s_S is rewritten by the next iteration's sums)
}

// EBM: 0 the triangle walk, 1 edge blocks in registers (every row of the mesh a fan of at most six blocks: structured meshes),
// 2 edge blocks with OVERFLOW (round 4: rows that are one fan of any length -- what gmsh's frontal meshes look like, a quarter
// of their nodes with seven neighbours: blocks beyond the six in registers sit in an LDS pool of 32-byte records).
// ONE: the single-workgroup instantiation (the mesh is at most four tiles: persist_single_workgroup).  An instantiation of its
// own, not a branch: the edge-block kernel sits at 256 of 256 registers, and a conditional exchange made the allocator spill 24.
// NPTX: nodes per lane when not the shape's four -- 1: ONE TILE PER WORKGROUP, what a mesh of at most 256 tiles runs as (config 2:
// 99 tiles).  The general instantiation carries three dead node slots through every loop there, and their registers (the
// blocks alone are 36 per slot) are what puts it at the 256-register limit.
template <int B, bool MG, int THREADS, int EBM, bool ONE = false, int NPTX = 0>
__global__ void __launch_bounds__(THREADS) k_cg_persist(const PersistParams P)
{
    constexpr int kPersistThreads = THREADS;
    constexpr int NPT = NPTX ? NPTX : persist_npt(THREADS);
    constexpr bool EB = EBM != 0, OV = EBM == 2;
    constexpr bool OPQ = MAG_PERSIST_OPAQUE == 2 || (MAG_PERSIST_OPAQUE == 1 && OV); // see persist_block_sum
    constexpr int SB = NPT * THREADS / B - 1; // bias of the workgroup-relative ring entries, in tile images (see the remap below)
    extern __shared__ __attribute__((aligned(16))) double2 smem[];
    const int tid = threadIdx.x;
    const int cap = P.cap, maxh = P.maxh;
    // LDS per local tile l: coordinates[capx], p image[cap] (owned part = the CG vector p itself; its halo part holds the halo
    // nodes' p), halo r[maxh], x[B].  Registers per node: r, q, the ring words, the triangle weights.
    // (capx = cap; with overflow blocks the first area only ever holds q of the B owned nodes -- no coordinates on the chip --
    // and shrinks to B: the LDS it frees is the pool's)
    const int capx = OV ? B : cap;
    const int tile_words = capx + cap + maxh + B;
    double2 *s_rec = smem + (size_t)(NPT * THREADS / B) * tile_words; // 2 * grid pieces of the partial records
    double *s_red = (double *)(s_rec + 2 * 256);
    double *s_S = s_red + 4 * (kPersistThreads / 64);
    double *s_chunk = s_S + 4;
    double *s_part = s_chunk + 4 * 32; // persist_block_sum: the rows' partials of the four sums
    // overflow records (OV): two double2 each; several ranks: behind the interface words (s_opk, below)
    [[maybe_unused]] double2 *s_pool = (double2 *)(s_part + kPersistPartDoubles) + (MG ? NPT * THREADS / 4 : 0);
    // slot s of this lane: local node s * THREADS + tid, in local tile (s * THREADS + tid) / B (a scalar: wave-uniform)
    auto t_loc = [&](int s) { return __builtin_amdgcn_readfirstlane((s * THREADS + tid) / B); };
    auto t_lt = [&](int s) { return (s * THREADS + tid) % B; };
    auto t_xy = [&](int s) { return smem + (size_t)t_loc(s) * tile_words; };

    const int cgrid = (int)gridDim.x - (MG ? P.comm_wg : 0); // compute workgroups (an exchange workgroup may follow them)
    if (MG && P.comm_wg && (int)blockIdx.x == cgrid) {
        persist_comm_loop(P, s_S, s_rec);
        return;
    }
    constexpr int NH = kPersistNh; // halo entries per thread: the workgroup's halo nodes are dealt out over ALL threads
    const unsigned tag0 = (MG ? P.tag_base : 0u) + 1u; // tag of epoch e: tag0 + e - 1
    int32_t deg[NPT], ent[NPT];
    // multi-GPU: a node other ranks read has an interface slot and a mask of reading ranks (bit r).  Both live in LDS, one
    // word per node (slot | readers << 24; s_opk), flagged by bit 6 of the node's flags: as registers they were spilled, and
    // their reload inside the node slot waited for the memory counter -- i.e. for the slot's own granule stores.
    [[maybe_unused]] uint32_t *s_opk = (uint32_t *)(s_part + kPersistPartDoubles);
    // global (Hilbert) id of slot s's node: the workgroup's tiles are consecutive
    const int32_t node_base = __builtin_amdgcn_readfirstlane(((MG ? P.t0 : 0) + (int32_t)blockIdx.x * P.tiles_per_wg) * B);
    auto node_of = [&](int s) { return node_base + s * THREADS + tid; };
    int32_t hg[NH], hloc[NH]; // global id (-1: none) and LDS position (tile * tile_words-relative) of a halo entry
    PackedFlags<NPT> flags; // per node slot: bit 0/1 prescribed ux/uy, 2 published, 3 live tile, 4 valid node, 5 fan closed in the blocks
    // overflow blocks (OV): per slot (first record in the workgroup's pool: 12 bits | records: 4 bits), two slots per register;
    // and, as scalars, the most records any lane of this wave has per slot (the overflow loop's trip count)
    [[maybe_unused]] uint32_t ovpk[(NPT + 1) / 2] = {};
    [[maybe_unused]] int32_t ovmax[NPT] = {};
    uint32_t w[NPT][kPersistRegs];
    int64_t ell_off[NPT];
    double2 r[NPT], q[NPT];

    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < NPT; ++s) {
        const int l = t_loc(s), lt = t_lt(s);
        const int32_t t = (MG ? P.t0 : 0) + blockIdx.x * P.tiles_per_wg + l;
        double2 *xy = t_xy(s), *pim = xy + capx, *hr = pim + cap, *xs = hr + maxh;
        if (MG) s_opk[s * THREADS + tid] = 0xffffffffu;
        deg[s] = 0;
        ent[s] = 0;
        flags.set(s, 3);
        ell_off[s] = 0;
        r[s] = q[s] = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < kPersistRegs; ++k) w[s][k] = 0xffffffffu;
        if (!(l < P.tiles_per_wg && t < (MG ? P.t1 : P.T))) continue;
        const TileMeta tm = P.meta[t];
        const int64_t nd = (int64_t)t * B + lt;
        flags.set(s, 8);
        if (nd < P.N) {
            const double2 b = P.bP[nd];
            r[s] = make_double2(-b.x, -b.y);
            // edge blocks: the coordinates are never read; q lives in their place (and the blocks in q's registers)
            xy[lt] = EB ? make_double2(0.0, 0.0) : P.xyP[nd];
            const uint32_t mk = P.maskP[nd];
            flags.set(s, flags[s] | 16u | (mk & 7u));
            // bit 3 of the mask (k_mark_external): read through memory by a tile of ANOTHER workgroup, or by a sibling tile
            // that keeps its halo copies; a node only its siblings read through their LDS slots publishes nothing
            if (kPersistSiblings && !(mk & 8u)) flags.set(s, flags[s] & ~4u);
            acc[0] = fma(b.y, b.y, fma(b.x, b.x, acc[0]));
        } else {
            xy[lt] = make_double2(0.0, 0.0);
            flags.set(s, flags[s] | 3u);
        }
        pim[lt] = make_double2(0.0, 0.0);
        xs[lt] = make_double2(0.0, 0.0);
        deg[s] = tm.deg;
#if MAG_PERSIST_ENT_SCALAR
        ent[s] = __builtin_amdgcn_readfirstlane(tm.ent); // one tile per wave and slot
#else
        ent[s] = tm.ent;
#endif
        ell_off[s] = tm.ell_off + lt;
#pragma unroll
        for (int k = 0; k < kPersistRegs; ++k)
            if (k < deg[s]) w[s][k] = P.ell16[ell_off[s] + (int64_t)k * B];
        // Ring entries become slots RELATIVE TO THE WORKGROUP'S TILES: entry = SB * tile_words + (owner tile - this tile) *
        // tile_words + position, read through this tile's base moved down by SB * tile_words; SB = tiles per workgroup - 1 (3
        // with 512-node tiles, 7 with 256-node ones: the first tile's references reach SB tiles up, the last one's SB down;
        // 2 SB + 1 tile images fit the 15 bits whenever the tiles fit the LDS).
        // A reference to a node that a SIBLING tile of this workgroup owns then points straight at the owner's slots (its
        // coordinates, its p), and the reader keeps no halo copy of it: no q to fetch for it, no r / p to advance -- about half
        // of a workgroup's halo entries when its tiles are consecutive in the Hilbert order.  Only for tiles whose rows all sit
        // in the registers (longer rows read tile-local entries from memory every iteration and keep their halo copies).
        // (One base per tile rather than one for the workgroup: with a common base ROCm 7.2's allocator spills 30 registers
        // into the walks.)
        {
            const int32_t t_first = (MG ? P.t0 : 0) + blockIdx.x * P.tiles_per_wg, t_end = MG ? P.t1 : P.T;
            // (several ranks: siblings are tiles of the same workgroup, hence of the same rank; rehearsed with two ranks
            // sharing one GPU at four tiles per workgroup, scripts/mg_share_ab.sh)
            // (OV: every entry, the pool's included, is rewritten here at start-up -- rows of any length)
            const bool short_rows = kPersistSiblings && (OV || tm.ent <= 2 * kPersistRegs);
            auto remap = [&](uint32_t e) -> uint32_t {
                if (e == 0xffffu) return e;
                const uint32_t lid = e & 0xfffu;
                uint32_t slot = (uint32_t)(SB * tile_words) + lid;
                if (short_rows && lid >= (uint32_t)B) {
                    const int32_t g = P.halo_g[tm.hoff + (int32_t)(lid - B)];
                    const int32_t ot = g / B, ol = ot - t_first;
                    if (ol >= 0 && ol < P.tiles_per_wg && ot < t_end) slot = (uint32_t)((SB + ol - l) * tile_words + (g - ot * B));
                }
                return slot | (e & 0x8000u);
            };
#pragma unroll
            for (int k = 0; k < kPersistRegs; ++k)
                if (k < deg[s]) w[s][k] = remap(w[s][k] & 0xffffu) | (remap(w[s][k] >> 16) << 16);
            if (OV) {
                // this node's blocks beyond the registers: k_edge_blocks_ovf left them at ovf_off[node] + j with the ring
                // entry they multiply (tile-local); they move into the workgroup's pool with that entry rewritten like the others
                // (record 0 of the pool is a zero block on slot `SB * tile_words` -- local node 0 of the reading tile --: what a
                // lane with fewer records than its wave's longest row reads in the steps it has nothing for)
                int32_t cnt = 0;
                uint32_t off = 0;
                if (tid == 0 && s == 0) {
                    s_pool[0] = make_double2(0.0, 0.0);
                    s_pool[1] = make_double2(0.0, __hiloint2double(0, SB * tile_words));
                }
                if (nd < P.N) {
                    constexpr int NBk = kPersistBlockEntries;
                    const uint32_t info = P.row_info[nd];
                    const int32_t n = (int32_t)(info & 63u), nblk = (info & 0x40u) ? n - 1 : n;
                    if (info & 0x40u) flags.set(s, flags[s] | 32u); // a closed fan: its closing triangle is folded into the blocks
                    const int32_t g0 = P.ovf_off[nd];
                    cnt = nblk > NBk ? nblk - NBk : 0;
                    off = 1u + (uint32_t)(g0 - P.ovf_off[(int64_t)t_first * B]);
                    const double2 *src = (const double2 *)P.ovf_rec + 2 * (int64_t)g0;
                    for (int32_t k = 0; k < cnt; ++k) {
                        const double2 a = src[2 * k];
                        double2 b = src[2 * k + 1];
                        b.y = __hiloint2double(0, (int)(remap((uint32_t)__double2loint(b.y) & 0xfffu) & 0x7fffu));
                        s_pool[2 * (off + (uint32_t)k)] = a;
                        s_pool[2 * (off + (uint32_t)k) + 1] = b;
                    }
                    // an OPEN fan of more than NB entries: its last entry takes the place of register entry NB - 1 (the walk
                    // telescopes the antisymmetric parts to u_last - u_first from the registers); k_edge_blocks_ovf has put the
                    // last block there and the middle ones into the pool
                    if (!(info & 0x40u) && n > NBk) {
                        const uint32_t ww = P.ell16[ell_off[s] + (int64_t)((n - 1) >> 1) * B];
                        const uint32_t last = remap(((n - 1) & 1) ? (ww >> 16) : (ww & 0xffffu)) & 0x7fffu;
                        constexpr int kw = (NBk - 1) >> 1;
                        w[s][kw] = ((NBk - 1) & 1) ? ((w[s][kw] & 0xffffu) | (last << 16)) : ((w[s][kw] & 0xffff0000u) | last);
                    }
                }
                ovpk[s >> 1] |= ((off & 0xfffu) | ((uint32_t)cnt << 12)) << (16 * (s & 1));
                int32_t m = 0;
                for (int32_t c = 1; c <= 15; ++c)
                    if (__any(cnt >= c ? 1 : 0)) m = c;
                ovmax[s] = __builtin_amdgcn_readfirstlane(m);
            }
        }
        if (EB) {
            // the fan closes inside the blocks when entry NB is entry 0's node again (k_edge_blocks folded that triangle in);
            // entries the tile's rows do not reach repeat the last one, so the walk gathers all NB without a test
            constexpr int NBk = kPersistBlockEntries;
            static_assert(NBk + 1 <= 2 * kPersistRegs, "entry NB is looked at in the registers");
            const uint32_t e0 = w[s][0] & 0xffffu, eN = (NBk & 1) ? (w[s][NBk >> 1] >> 16) : (w[s][NBk >> 1] & 0xffffu);
            if (!OV && tm.ent > NBk && !(eN & 0x8000u) && (eN & 0x7fffu) == (e0 & 0x7fffu)) flags.set(s, flags[s] | 32u);
            ring_pad_entries<kPersistRegs, NBk, 0x7fffu>(w[s], tm.ent);
        }
        if ((flags[s] & 20) == 20) put_granules(P.qg + 4 * nd, tag0, make_double2(0.0, 0.0)); // q_{-1} = 0, parity 0
        if (MG && (flags[s] & 16)) {
            const int32_t osl = P.own_qslot[nd];
            if (osl >= 0) {
                uint32_t ord = 0;
                if (!P.win_shared) ord = P.iface_readers[osl];
                asm volatile("" : "+v"(ord)); // landed: nothing of it is in flight when the stores begin
                s_opk[s * THREADS + tid] = (uint32_t)osl | (ord << 24); // (an inbox holds far fewer than 2^24 slots: the host checks)
                flags.set(s, flags[s] | 64u);
                publish_q(P, 0, osl, ord, tag0, make_double2(0.0, 0.0));
            }
        }
    }
    // halo entries of the workgroup's tiles, in tile order, dealt out round-robin: thread t takes entries t, t + 512
    if (kPersistSiblings) {
        // About half of the entries are owned by sibling tiles and need no halo copy.  The ones that do are
        // COMPACTED before they are dealt out (same order): ~250 of them fill the first round of four waves, where the
        // uncompacted list left a few live lanes in both rounds of all eight -- every one of those wave-rounds pays the LDS
        // operations of the halo update and the instructions of the halo fetch in every iteration.
        const int32_t t_first = (MG ? P.t0 : 0) + blockIdx.x * P.tiles_per_wg, t_end = MG ? P.t1 : P.T;
        int32_t cg[NH], cl[NH];
#pragma unroll
        for (int e = 0; e < NH; ++e) {
            cg[e] = -1;
            cl[e] = 0;
            int32_t rem = tid + kPersistThreads * e;
            for (int l = 0; l < P.tiles_per_wg && t_first + l < t_end; ++l) {
                const TileMeta tm = P.meta[t_first + l];
                if (rem < tm.nh) {
                    const int32_t g = P.halo_g[tm.hoff + rem];
                    const int32_t ot = g / B, ol = ot - t_first;
                    // a sibling tile owns it: this tile's walks read the owner's slots (rows all in the registers only)
                    const bool sibling = (OV || tm.ent <= 2 * kPersistRegs) && ol >= 0 && ol < P.tiles_per_wg && ot < t_end;
                    if (!sibling) {
                        cg[e] = g;
                        cl[e] = l * tile_words + rem;
                    }
                    break;
                }
                rem -= tm.nh;
            }
        }
        int32_t *s_cnt = (int32_t *)s_red; // live entries per (round, wave), then their exclusive prefix
        int2 *s_list = (int2 *)s_rec;      // 2 * THREADS entries at most: 8 KB, the record staging area (free until the exchange)
        constexpr int NW8 = kPersistThreads / 64;
        int32_t pos[NH];
#pragma unroll
        for (int e = 0; e < NH; ++e) {
            const unsigned long long live = __ballot(cg[e] >= 0);
            pos[e] = __builtin_amdgcn_mbcnt_hi((unsigned)(live >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)live, 0u));
            if ((tid & 63) == 0) s_cnt[e * NW8 + (tid >> 6)] = __popcll(live);
        }
        __syncthreads();
        int32_t total = 0;
#pragma unroll
        for (int e = 0; e < NH; ++e) {
            int32_t base = 0;
            for (int k = 0; k < e * NW8 + (tid >> 6); ++k) base += s_cnt[k];
            if (cg[e] >= 0) s_list[base + pos[e]] = make_int2(cg[e], cl[e]);
        }
        for (int k = 0; k < NH * NW8; ++k) total += s_cnt[k];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NH; ++e) {
            hg[e] = -1;
            hloc[e] = 0;
            const int32_t c = tid + kPersistThreads * e;
            if (c < total) {
                const int2 ent2 = s_list[c];
                hg[e] = ent2.x;
                hloc[e] = ent2.y;
                const int32_t l = ent2.y / tile_words, rem = ent2.y - l * tile_words;
                const TileMeta tm = P.meta[t_first + l];
                double2 *xy = smem + (size_t)l * tile_words;
                if (!EB) xy[B + rem] = P.halo_xy[tm.hoff + rem];
                const double2 hb = P.bP[hg[e]];
                xy[capx + cap + rem] = make_double2(-hb.x, -hb.y); // halo r
                xy[capx + B + rem] = make_double2(0.0, 0.0);       // halo p: its slot in the p image
                if (MG) { // a node another rank owns: its q comes through the window (slot s encoded as -2 - s)
                    const int32_t hs = P.halo_qslot[tm.hoff + rem];
                    if (hs >= 0) hg[e] = -2 - hs;
                }
            }
        }
        __syncthreads(); // s_red and s_rec go back to their day jobs
    } else {
        const int32_t t_first = (MG ? P.t0 : 0) + blockIdx.x * P.tiles_per_wg, t_end = MG ? P.t1 : P.T;
#pragma unroll
        for (int e = 0; e < NH; ++e) {
            hg[e] = -1;
            hloc[e] = 0;
            int32_t rem = tid + kPersistThreads * e;
            for (int l = 0; l < P.tiles_per_wg && t_first + l < t_end; ++l) {
                const TileMeta tm = P.meta[t_first + l];
                if (rem < tm.nh) {
                    double2 *xy = smem + (size_t)l * tile_words;
                    hg[e] = P.halo_g[tm.hoff + rem];
                    hloc[e] = l * tile_words + rem;
                    if (!EB) xy[B + rem] = P.halo_xy[tm.hoff + rem];
                    const double2 hb = P.bP[hg[e]];
                    xy[capx + cap + rem] = make_double2(-hb.x, -hb.y);   // halo r
                    xy[capx + B + rem] = make_double2(0.0, 0.0);         // halo p: its slot in the p image
                    if (MG) { // a node another rank owns: its q comes through the window (slot s encoded as -2 - s)
                        const int32_t hs = P.halo_qslot[tm.hoff + rem];
                        if (hs >= 0) hg[e] = -2 - hs;
                    }
                    break;
                }
                rem -= tm.nh;
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0) acc[1] = 1.0; // "p.q" > 0: alpha finite, multiplies q = 0
    persist_block_sum<THREADS, OPQ>(acc, s_red, s_part);
    int par = 0;
    unsigned epoch = tag0; // the tags of successive exchanges
    constexpr bool single_wg = ONE && !MG; // the whole mesh in this workgroup: no exchange at all
    if (tid < 2) // the block sums are in every thread: two threads publish the record's two pieces
        put_granules(P.recg + 4 * (2 * ((int64_t)par * cgrid + blockIdx.x) + tid), epoch,
                     tid == 0 ? make_double2(acc[0], acc[1]) : make_double2(acc[2], acc[3]));
    if (single_wg && tid < 2) ((double2 *)s_S)[tid] = tid == 0 ? make_double2(acc[0], acc[1]) : make_double2(acc[2], acc[3]);
    double2 hq[NH]; // q of this thread's halo nodes
    double Sx[4] = {0.0, 0.0, 0.0, 0.0}; // the four grid-wide sums as the single-GPU exchange hands them over (scalars)
    if (single_wg)
        persist_single_workgroup<NH>(s_S, Sx, hq);
    else if (MG ? !persist_exchange_mg<NH>(P, par, epoch, hg, hq, s_S, s_rec)
                : !persist_exchange<NH, EB, OPQ>(P, par, epoch, hg, hq, s_S, s_rec, s_chunk, s_part, Sx))
        return;

    const double c0 = P.c0, nu = P.nu, h = P.h;
    // iteration-invariant part of the ring walks: the triangles' weights c0 / (2A), once per solve (the exchange above
    // ended with a workgroup barrier: the coordinates are staged)
#ifndef MAG_PERSIST_WEIGHTS_MG
#define MAG_PERSIST_WEIGHTS_MG 6
#endif
    constexpr int kW = MG ? MAG_PERSIST_WEIGHTS_MG : kPersistWeights;
    constexpr int NCW = kW > 0 ? kW : 1;
    // Edge blocks (EB instantiation; cg_device.h, ring_blocks): every node's triangles folded into NB symmetric 2 x 2 blocks,
    // once per solve; a ring step is then one gather of p and six fp64 operations, and the coordinates are not read again.
    // Only for meshes whose rows are ALL one fan of at most NB entries, or NB + 1 with the last one closing onto the first
    // (k_ring16 says so: closed fans of valence <= NB, open ones of <= NB - 1 triangles -- every structured mesh); the host
    // launches the triangle-walk instantiation for any other mesh.
    constexpr int NB = kPersistBlockEntries;
    constexpr bool BLOCKS = EB;
    constexpr bool CACHED = !BLOCKS && kW > 0;
    constexpr int NKB = BLOCKS ? 3 * NB : NCW;
    double wgt[NPT][NKB];
    const double kappa = uniform_f64(0.5 * (h - nu) * c0);
    if (BLOCKS) { // computed by k_edge_blocks before the launch: loads, no coordinates on the chip at all
#pragma unroll
        for (int s = 0; s < NPT; ++s) {
            const bool live = (flags[s] & 24) == 24;
#pragma unroll
            for (int c = 0; c < NKB; ++c) wgt[s][c] = live ? P.kblocks[(int64_t)c * P.kb_stride + node_of(s)] : 0.0;
        }
        // Every load has landed before the loop: a block still "in flight" at the loop's entry makes the compiler wait for
        // the memory counter inside every node slot's walk -- where that counter also holds the previous slot's granule
        // stores, i.e. a full store round trip per slot and iteration.
#pragma unroll
        for (int s = 0; s < NPT; ++s)
#pragma unroll
            for (int c = 0; c < NKB; ++c) asm volatile("" : "+v"(wgt[s][c]));
    } else if (CACHED) {
#pragma unroll
        for (int s = 0; s < NPT; ++s) {
#pragma unroll
            for (int c = 0; c < NKB; ++c) wgt[s][c] = 0.0;
            if (!(flags[s] & 8)) continue;
            const double2 *xy = t_xy(s);
            const int32_t nent = __builtin_amdgcn_readfirstlane(ent[s]);
            if (nent > 0)
                ring_weights<kPersistRegs, NCW, 0x7fffu>(w[s], nent, xy - SB * tile_words, xy[t_lt(s)], c0,
                                                         *reinterpret_cast<double(*)[NCW]>(&wgt[s][0]));
        }
    }
    // Edge-block instantiation: q in LDS, in the coordinates' place, instead of registers (the blocks need them).  q rather
    // than r: q is written once (end of the node's walk) and read once (r += alpha q) per iteration, r is read in the
    // update, the dots and the deferred x update and written in the update -- two LDS operations per node and iteration
    // instead of four.
    constexpr bool QL = EB;
    bool all_live = true; // every node slot of this workgroup carries a live tile (a scalar)
#pragma unroll
    for (int s = 0; s < NPT; ++s) all_live = all_live && (flags[s] & 8) != 0;
    all_live = __builtin_amdgcn_readfirstlane(__syncthreads_and(all_live ? 1 : 0)) != 0;
    double target = P.tol, bb = 0.0;
    long long j = 0;
    int verdict = 0; // 1 converged, 2 iteration cap, 3 non-finite
    double cost = 0.0;
    // argmin's best_param bookkeeping (solver.rs:167-174): lowest cost so far and the iteration that had it, kept by
    // the one thread that reports the verdict, in LDS (no registers of the other 511 lanes)
    const bool reporter = blockIdx.x == 0 && tid == 0;
    if (reporter) {
        s_chunk[0] = __builtin_inf();
        ((long long *)s_chunk)[1] = 0;
    }
#ifdef MAG_PERSIST_STAMPS
    // the stamping lane's accumulators live in LDS (the spare words of s_chunk): the edge-block instantiation has no registers
    // for them, and a spilled register in the loop would be measured along with the phases
    unsigned long long *stamp_sum = (unsigned long long *)(s_chunk + 16); // [kStampPhases], then `last`, then the count, then scratch
    unsigned long long &stamp_last = stamp_sum[kStampPhases], &stamp_iters = stamp_sum[kStampPhases + 1];
    if (tid == MAG_PERSIST_STAMP_TID)
        for (int k = 0; k < kStampPhases + 2; ++k) stamp_sum[k] = 0;
#endif
    for (;;) {
#ifdef MAG_PERSIST_STAMPS
        const bool stamping = tid == MAG_PERSIST_STAMP_TID && j >= kStampFrom && j < kStampTo;
        if (stamping) {
            stamp_last = __builtin_amdgcn_s_memrealtime();
            ++stamp_iters;
        }
#endif
        // (single GPU: the exchange hands the sums over in scalar registers already)
        const double S0 = MG ? uniform_f64(s_S[0]) : Sx[0], S1 = MG ? uniform_f64(s_S[1]) : Sx[1],
                     S2 = MG ? uniform_f64(s_S[2]) : Sx[2], S3 = MG ? uniform_f64(s_S[3]) : Sx[3];
        if (j == 0) {
            bb = S0;
            target = uniform_f64(P.stop_mode == 2 ? P.tol * sqrt(bb) : P.tol);
        }
        const double rr = S0;
        // alpha and beta before the stop test: their two division chains (~300 cycles, every wave, every iteration) then run
        // alongside the square root of the cost instead of behind it; on the way out they are simply not used
        const double alpha = uniform_f64(rr / S1);
        const double beta = uniform_f64(fma(alpha * alpha, S3, fma(2.0 * alpha, S2, rr)) / rr);
        cost = uniform_f64(P.stop_mode == 1 ? fabs(rr) : sqrt(rr));
        const long long it_done = j - 1;
        if (reporter && it_done >= 1 && it_done - 1 < P.hist_len) P.hist[it_done - 1] = cost;
        if (j == 0 && bb == 0.0) {
            verdict = 1;
            cost = 0.0;
            break;
        }
        if (reporter && it_done >= 1 && cost < s_chunk[0]) {
            s_chunk[0] = cost;
            ((long long *)s_chunk)[1] = it_done;
        }
        if (it_done >= 1 && cost <= target) verdict = 1;
        else if (!(fabs(rr) <= 1.79769313486231570e308)) verdict = 3;
        else if (it_done >= P.max_iter) verdict = 2;
        if (verdict) break;

        // x += alpha p is the one update nothing in the iteration waits for.  It is done in the idle time before the first
        // sweep of the exchange, from what is on the chip by then: alpha p_{j-1} = (alpha / beta) (p_j + r_j) (p_j = -r_j +
        // beta p_{j-1}); x feeds back into nothing, so the iterates and the iteration count are untouched and x itself moves by
        // a few ulps per step.  beta = 0 (an exactly zero residual) cannot be divided by: then, and only then, here.
        const bool xnow = !(beta != 0.0) || !(fabs(beta) <= 1.79769313486231570e308);
        // ---- vector updates: r in registers, p and x in LDS, halo copies in LDS (their q from the publishers)
        if (!kPersistDeferX || xnow) { // x += alpha p here, on the critical path, only when it cannot be rebuilt later
#pragma unroll
            for (int s = 0; s < NPT; ++s) {
                if (!(flags[s] & 8)) continue;
                const int lt = t_lt(s);
                double2 *xy = t_xy(s), *pim = xy + capx, *xs = pim + cap + maxh;
                const double2 po = pim[lt];
                double2 xo = xs[lt];
                xo.x = fma(alpha, po.x, xo.x);
                xo.y = fma(alpha, po.y, xo.y);
                xs[lt] = xo;
            }
        }
        auto update_slot = [&](int s) {
            const int lt = t_lt(s);
            double2 *xy = t_xy(s), *pim = xy + capx;
            const double2 po = pim[lt];
            const double2 qv = QL ? xy[lt] : q[s];
            double2 pn;
            // (explicit FMAs: a halo copy of this node in another workgroup runs the same recurrence and must get the same bits)
            r[s].x = fma(alpha, qv.x, r[s].x);
            r[s].y = fma(alpha, qv.y, r[s].y);
            pn.x = fma(beta, po.x, -r[s].x);
            pn.y = fma(beta, po.y, -r[s].y);
            pim[lt] = pn;
        };
        // A workgroup whose node slots are all live (every one on the 1M mesh but the last) runs them as ONE basic block: the
        // eight LDS loads of the four slots go out together.  With a test per slot each slot waited for its own two loads
        // before the next slot's were issued.  (A dead slot's LDS is never read by anybody else: it is skipped, not zeroed.)
        if (all_live) {
#pragma unroll
            for (int s = 0; s < NPT; ++s) update_slot(s);
        } else {
#pragma unroll
            for (int s = 0; s < NPT; ++s)
                if (flags[s] & 8) update_slot(s);
        }
#pragma unroll
        for (int e = 0; e < NH; ++e)
            if (hg[e] != -1) {
                double2 *hbase = smem + hloc[e]; // = tile base + position: coordinates at [B], p image at [cap + B], ...
                double2 hrv = hbase[capx + cap], hpv = hbase[capx + B]; // the halo node's p lives in the p image itself
                hrv.x = fma(alpha, hq[e].x, hrv.x);
                hrv.y = fma(alpha, hq[e].y, hrv.y);
                hpv.x = fma(beta, hpv.x, -hrv.x);
                hpv.y = fma(beta, hpv.y, -hrv.y);
                hbase[capx + cap] = hrv;
                hbase[capx + B] = hpv;
            }
        MAG_STAMP(0) // scalars + vector updates issued
        __syncthreads();
        MAG_STAMP(1) // ... landed in LDS for everybody (workgroup barrier)

        // ---- q = M K M p on the owned nodes, dot partials, publication
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = 0.0;
        // the ring words stay packed: unpacked once and for all (loop-invariant) they take a register per entry -- 24 in the
        // edge-block instantiation, 40 in the others (where MAG_PERSIST_PACK_MG does the same for the multi-GPU instantiation:
        // the registers go to the cached triangle weights)
        if ((EB && MAG_PERSIST_EB_PACK) || (MG && !EB && MAG_PERSIST_PACK_MG)) {
#pragma unroll
            for (int s = 0; s < NPT; ++s)
#pragma unroll
                for (int k = 0; k < (EB ? (kPersistBlockEntries + 1) / 2 : kPersistRegs); ++k) asm volatile("" : "+v"(w[s][k]));
        }
        if (OV) { // ... and so do the pool positions: unpacked outside the loop they are eight registers, spilled
#pragma unroll
            for (int i = 0; i < (NPT + 1) / 2; ++i) asm volatile("" : "+v"(ovpk[i]));
        }
#pragma unroll
        for (int s = 0; s < NPT; ++s) {
#if MAG_PERSIST_PRIO == 3
            // The two waves of a SIMD run the same program, and at equal priority the older one (waves 0-3) wins every
            // arbitration: the stamps showed it through its walks 2 us before its partner, which then ran alone, at
            // single-wave efficiency.  Waves 4-7 take priority 1 for their LAST TWO node slots (back to 0 before the sums):
            // the partners finish within 0.8 us of each other.  Twelve other schedules (static priority, alternating per
            // slot or pair, one or three slots, three levels, priority during the vector updates) were measured and are in
            // profiles/r03_persist_ab.txt (q0-q16): none better.
            if (((tid >> 8) & 1) && s >= NPT / 2) __builtin_amdgcn_s_setprio(1);
#endif
#ifndef MAG_PERSIST_OVF_PRIO
#define MAG_PERSIST_OVF_PRIO 2 // frontal1m: 6.39 (off), 6.31 (1), 6.30 (2) us per iteration
#endif
#if MAG_PERSIST_OVF_PRIO
            // (OV) a wave that carries this slot's long rows -- the valence partition gives every wave one such slot --
            // takes priority for it: it is the one the workgroup's sums will wait for
            if (OV && ovmax[s] > 0) __builtin_amdgcn_s_setprio(MAG_PERSIST_OVF_PRIO);
            else if (OV && !(((tid >> 8) & 1) && s >= NPT / 2)) __builtin_amdgcn_s_setprio(0);
#endif
            if (!(flags[s] & 8)) continue;
            const int lt = t_lt(s);
            const double2 *xy = t_xy(s), *pim = xy + capx;
            const double2 ca = xy[lt], pa = pim[lt];
            double fx = 0.0, fy = 0.0;
            {
                const int32_t nent = __builtin_amdgcn_readfirstlane(ent[s]); // one tile per wave: a scalar
                if (nent > 0) { // entries are biased slots relative to this tile (see the remap at the top)
                    const uint32_t toff = (uint32_t)(SB * tile_words);
                    if (OV) {
                        const uint32_t oc = (ovpk[s >> 1] >> (16 * (s & 1))) & 0xffffu;
                        ring_walk_blocks_ovf<kPersistRegs, NB, 0x7fffu>(w[s], pim - SB * tile_words, pa, kappa, (flags[s] & 32u) != 0,
                                                                        *reinterpret_cast<const double(*)[3 * NB]>(&wgt[s][0]), s_pool,
                                                                        oc & 0xfffu, oc >> 12, ovmax[s], fx, fy);
                    } else if (BLOCKS)
                        ring_walk_blocks<kPersistRegs, NB, 0x7fffu>(w[s], pim - SB * tile_words, pa, kappa, (flags[s] & 32u) != 0,
                                                                    *reinterpret_cast<const double(*)[3 * NB]>(&wgt[s][0]), fx, fy);
                    else if (CACHED)
                        ring_walk_cached<kPersistRegs, NCW, 0x7fffu>(w[s], P.ell16 + ell_off[s], B, nent, xy - SB * tile_words, pim - SB * tile_words,
                                                                     ca, pa, c0, nu, h, *reinterpret_cast<const double(*)[NCW]>(&wgt[s][0]),
                                                                     fx, fy, toff);
                    else
                        ring_walk_uniform<kPersistRegs, kPersistBlock, kPersistBlock2, 0x7fffu>(
                            w[s], P.ell16 + ell_off[s], B, nent, xy - SB * tile_words, pim - SB * tile_words, ca, pa, c0, nu, h, fx, fy,
                            toff);
                }
            }
            if ((flags[s] & 1) || !(flags[s] & 16)) fx = 0.0;
            if ((flags[s] & 2) || !(flags[s] & 16)) fy = 0.0;
            const double2 qn = make_double2(fx, fy);
            if (QL) t_xy(s)[lt] = qn;
            else q[s] = qn;
#if MAG_PERSIST_SADDR
            if ((flags[s] & 20) == 20) // (2 x 32 N bytes of granules: below 4 GB for every mesh the chip can hold)
                put_granules_at(P.qg, 32u * ((uint32_t)(par ^ 1) * (uint32_t)P.N + (uint32_t)node_of(s)), epoch + 1, qn);
#else
            if ((flags[s] & 20) == 20) put_granules(P.qg + 4 * ((int64_t)(par ^ 1) * P.N + node_of(s)), epoch + 1, qn);
#endif
            if (MG && (flags[s] & 64u)) {
                const uint32_t opk = s_opk[s * THREADS + tid];
                publish_q(P, par ^ 1, (int32_t)(opk & 0xffffffu), opk >> 24, epoch + 1, qn);
            }
            acc[0] = fma(r[s].y, r[s].y, fma(r[s].x, r[s].x, acc[0]));
            acc[1] = fma(pa.y, fy, fma(pa.x, fx, acc[1]));
            acc[2] = fma(r[s].y, fy, fma(r[s].x, fx, acc[2]));
            acc[3] = fma(fy, fy, fma(fx, fx, acc[3]));
        }
#if MAG_PERSIST_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        MAG_STAMP(2) // ring walks of this wave's nodes, q published
#ifdef MAG_PERSIST_STAMPS
        unsigned long long *sub_ = stamp_sum + kStampPhases + 8;
        persist_block_sum<THREADS, OPQ>(acc, s_red, s_part, stamping ? sub_ : nullptr);
#else
        persist_block_sum<THREADS, OPQ>(acc, s_red, s_part);
#endif
        par ^= 1;
        ++epoch;
        ++j;
        if (tid < 2)
            put_granules(P.recg + 4 * (2 * ((int64_t)par * cgrid + blockIdx.x) + tid), epoch,
                         tid == 0 ? make_double2(acc[0], acc[1]) : make_double2(acc[2], acc[3]));
        if (single_wg && tid < 2) ((double2 *)s_S)[tid] = tid == 0 ? make_double2(acc[0], acc[1]) : make_double2(acc[2], acc[3]);
#ifdef MAG_PERSIST_STAMPS
        if (stamping) {
            const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();
            stamp_sum[8] += sub_[0] - stamp_last;
            stamp_sum[9] += sub_[1] - sub_[0];
            stamp_sum[10] += now_ - sub_[1];
            sub_[2] = now_;
        }
#endif
        if (kPersistDeferX && !xnow) { // the deferred x += alpha p_{j-1}, in the shadow of the exchange's first wait
            const double ab = uniform_f64(alpha / beta);
            auto x_slot = [&](int s) {
                const int lt = t_lt(s);
                double2 *xy = t_xy(s), *pim = xy + capx, *xs = pim + cap + maxh;
                const double2 pj = pim[lt], rv = r[s];
                double2 xo = xs[lt];
                xo.x = fma(ab, pj.x + rv.x, xo.x);
                xo.y = fma(ab, pj.y + rv.y, xo.y);
                xs[lt] = xo;
            };
            if (all_live) {
#pragma unroll
                for (int s = 0; s < NPT; ++s) x_slot(s);
            } else {
#pragma unroll
                for (int s = 0; s < NPT; ++s)
                    if (flags[s] & 8) x_slot(s);
            }
        }
#ifdef MAG_PERSIST_STAMPS
        if (stamping) stamp_sum[11] += __builtin_amdgcn_s_memrealtime() - sub_[2];
#endif
        MAG_STAMP(3) // workgroup sums (wave trees, barrier, eight waves in order) + record published
#ifdef MAG_PERSIST_STAMPS
        unsigned long long *xs_ = stamp_sum + kStampPhases + 2; // (LDS as well: no stack object, no scratch in the diagnostic build)
        if (stamping) xs_[0] = xs_[1] = xs_[2] = xs_[3] = xs_[4] = 0;
        if (single_wg)
            persist_single_workgroup<NH>(s_S, Sx, hq);
        else if (MG ? !persist_exchange_mg<NH>(P, par, epoch, hg, hq, s_S, s_rec, stamping ? xs_ : nullptr)
                    : !persist_exchange<NH, EB, OPQ>(P, par, epoch, hg, hq, s_S, s_rec, s_chunk, s_part, Sx, stamping ? xs_ : nullptr))
            return;
        if (stamping) { // inside the exchange: wait before the first sweep / sweeps until complete / record reduction
            stamp_sum[4] += xs_[0] - stamp_last;
            stamp_sum[5] += xs_[1] - xs_[0];
            const unsigned long long end_ = __builtin_amdgcn_s_memrealtime();
            stamp_sum[6] += end_ - xs_[1];
            stamp_sum[7] += xs_[2]; // sweeps taken
            stamp_sum[12] += xs_[3] - xs_[4];
            stamp_sum[13] += end_ - xs_[3];
        }
#else
        if (single_wg)
            persist_single_workgroup<NH>(s_S, Sx, hq);
        else if (MG ? !persist_exchange_mg<NH>(P, par, epoch, hg, hq, s_S, s_rec)
                    : !persist_exchange<NH, EB, OPQ>(P, par, epoch, hg, hq, s_S, s_rec, s_chunk, s_part, Sx))
            return;
#endif
    }
#ifdef MAG_PERSIST_STAMPS
    if (tid == MAG_PERSIST_STAMP_TID && P.stamps) {
        unsigned long long *o = P.stamps + (size_t)blockIdx.x * (kStampPhases + 1);
        for (int k = 0; k < kStampPhases; ++k) o[k] = stamp_sum[k];
        o[kStampPhases] = stamp_iters;
    }
#endif
    // x of iterate j-1 is in LDS; the verdict is the same in every workgroup
#pragma unroll
    for (int s = 0; s < NPT; ++s)
        if ((flags[s] & 24) == 24) P.x[node_of(s)] = (t_xy(s) + capx + cap + maxh)[t_lt(s)];
    if (blockIdx.x == 0 && tid == 0) {
        FusedState *st = P.st;
        st->bb = bb;
        st->target = target;
        st->iterations = j - 1 < 0 ? 0 : j - 1;
        st->final_cost = cost;
        st->converged = verdict == 1 ? 1 : 0;
        st->breakdown = verdict == 3 ? 1 : 0;
        st->best_cost = s_chunk[0];
        st->best_iter = ((long long *)s_chunk)[1];
        st->done = 1;
    }
}

// The four-slot edge-block instantiation of ONE GPU -- the kernel of BASELINE config 3 -- is compiled in a translation unit of its
// own (persist_k4.o: this file again with -DMAG_PERSIST_TU_K4, everything below left out) under the max-ilp scheduler: 5.43
// against 5.52 us per iteration at 1M triangles in one session, while every other instantiation is FASTER under iterative-ilp
// (one / two / three node slots 3.48 / 4.08 / 4.65 against 3.53 / 4.16 / 4.83; the overflow instantiation indifferent).  The
// scheduler is a per-translation-unit option, hence the second object.  Builds without -DMAG_PERSIST_SPLIT_K4 (the stamped
// build, scripts/build_variant.sh) keep the instantiation in this unit.
#ifndef MAG_PERSIST_SPLIT_K4
#define MAG_PERSIST_SPLIT_K4 0
#endif
void persist_launch_eb1_k4(const PersistParams &P, int32_t grid, size_t lds, hipStream_t s);
#ifdef MAG_PERSIST_TU_K4
void persist_launch_eb1_k4(const PersistParams &P, int32_t grid, size_t lds, hipStream_t s)
{
    k_cg_persist<512, false, 512, 1><<<grid, 512, lds, s>>>(P);
}
} // namespace magk
#else

// ========================================= inbox exchange for the STREAMING kernels ===
// Meshes the chips cannot hold (more than ~0.5M nodes per GPU: BASELINE config 5 on 8 GPUs) run one fused launch per CG
// iteration, and the ranks must trade [dot partials | q on the interface nodes] between launches.  One RCCL all-reduce
// does that in 20-30 us; this kernel does it through the same per-rank inboxes the on-chip kernel uses (device memory,
// mapped by the peers; tagged granules, polls in local HBM), in place on the buffer the iteration kernel just filled --
// a drop-in for the all-reduce, the iteration kernels are untouched:
//   * q of an interface node this rank owns goes to the inboxes of the ranks that read it (reader mask);
//   * block 0 adds the rank's G partial slots in a fixed order and stores the rank's four sums into every inbox;
//   * block 0 waits for the R sums, adds them in rank order (the same bits on every rank) and writes the total into slot 0
//     of each partial array, clearing the other slots: the next launch's sum over the slots is the total; every block
//     waits for the q of its share of the interface nodes this rank reads and writes them into the buffer; slots the
//     rank neither owns nor reads get 0.
// Epochs alternate between two parities of the inbox; nobody can be two exchanges ahead of a rank that has not finished
// reading (it would need that rank's next sums first).  Every wait is bounded: a rank that gives up raises
// FusedState::exchange_timeout (and the other ranks' timeout words) and the host falls back to the all-reduce.
struct StreamExchangeParams {
    double *buf;              // [4 x g_all partial slots | n_iface x double2 q]: the iteration kernel's output, in place
    int32_t g_all, n_iface, rank, nranks, own0, own1, fpar; // fpar: parity of the iteration launch that filled buf
    uint32_t spin_limit;
    const int32_t *iface;     // sorted Hilbert ids of the interface nodes
    const uint8_t *iface_readers;
    uint8_t *inbox[8];
    FusedState *st;
};

__global__ void __launch_bounds__(256) k_stream_exchange(const StreamExchangeParams P)
{
    __shared__ double s_red[4][4];
    __shared__ double2 s_recs[16];
    const int tid = threadIdx.x, R = P.nranks;
    if (P.st->done) return; // converged (or capped) earlier in this block: every rank takes the same exit
    // Exchange e follows iteration launch e - 1, which left its successor's number e in jslot[fpar ^ 1] (fused_step): the
    // epoch, the inbox parity and the tag come from device memory, identical on every rank (all ranks run the same
    // launches), and nothing in this kernel's arguments changes from one iteration to the next.
    const uint32_t epoch = (uint32_t)P.st->jslot[P.fpar ^ 1];
    const int32_t xpar = (int32_t)(epoch & 1u);
    const uint32_t xtag = P.st->exchange_tag_base + epoch;
    uint8_t *mine = P.inbox[P.rank];
    gu32 *wtmo = (gu32 *)mine;
    double2 *q = (double2 *)(P.buf + 4 * (size_t)P.g_all);
    const size_t qoff = 64 + 128 * (size_t)R;
    // (1) this rank's interface q to their readers
    for (int32_t k = blockIdx.x * 256 + tid; k < P.n_iface; k += gridDim.x * 256) {
        const int32_t g = P.iface[k];
        if (g < P.own0 || g >= P.own1) continue;
        uint32_t readers = P.iface_readers[k];
        double2 v = q[k];
        // both landed before the first store: a load still in flight at the loop's entry would put a wait for the memory
        // counter -- which also counts the stores -- at the top of every round: one store round trip across xGMI per reader
        asm volatile("" : "+v"(readers), "+v"(v.x), "+v"(v.y));
        for (int r = 0; r < R; ++r)
            if (r != P.rank && ((readers >> r) & 1u))
                put_granules_sys((unsigned long long *)(P.inbox[r] + qoff + 32 * ((size_t)xpar * P.n_iface + k)), xtag, v);
    }
    // (2) block 0: the rank's sums, fixed order (thread-strided partials, wave trees, the four waves in order)
    if (blockIdx.x == 0) {
        double S[4] = {0.0, 0.0, 0.0, 0.0};
        for (int i = tid; i < P.g_all; i += 256) {
#pragma unroll
            for (int c = 0; c < 4; ++c) S[c] += P.buf[(size_t)c * P.g_all + i];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) S[c] = wave_sum_dpp(S[c]);
        if ((tid & 63) == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) s_red[c][tid >> 6] = S[c];
        }
        __syncthreads();
        if (tid < 2 * R) {
            double T[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) T[c] = ((s_red[c][0] + s_red[c][1]) + s_red[c][2]) + s_red[c][3];
            put_granules_sys((unsigned long long *)(P.inbox[tid >> 1] + 64) + 4 * (2 * ((int64_t)xpar * R + P.rank) + (tid & 1)),
                             xtag, (tid & 1) == 0 ? make_double2(T[0], T[1]) : make_double2(T[2], T[3]));
        }
    }
    // (3) wait: the R sums (every block) and the q this rank reads (each block its share of the slots)
    const unsigned long long *wrec = (const unsigned long long *)(mine + 64) + 8 * (int64_t)xpar * R;
    const unsigned long long *wq = (const unsigned long long *)(mine + qoff) + 4 * (int64_t)xpar * P.n_iface;
    bool have_w = tid >= 2 * R || blockIdx.x != 0; // only block 0 turns the sums into the next launch's input
    int32_t k = blockIdx.x * 256 + tid; // one slot at a time per thread
    bool ok_all = false;
    for (unsigned spins = 0; spins < P.spin_limit; ++spins) {
        if (!have_w) {
            double2 v;
            have_w = get_granules_sys(wrec, 64u * (uint32_t)R, 32u * (uint32_t)tid, xtag, v);
            if (have_w) s_recs[tid] = v;
        }
        while (k < P.n_iface) { // advance over the slots that are settled; stop at the first one still on its way
            const int32_t g = P.iface[k];
            const bool owned = g >= P.own0 && g < P.own1;
            if (!owned) {
                if ((P.iface_readers[k] >> P.rank) & 1u) {
                    double2 v;
                    if (!get_granules_sys(wq, 32u * (uint32_t)P.n_iface, 32u * (uint32_t)k, xtag, v)) break;
                    q[k] = v;
                } else {
                    q[k] = make_double2(0.0, 0.0);
                }
            }
            k += gridDim.x * 256;
        }
        if (__syncthreads_and((have_w && k >= P.n_iface) ? 1 : 0)) {
            ok_all = true;
            break;
        }
        if ((spins & 255u) == 255u) {
            const int dead = tid == 0 && __hip_atomic_load(wtmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u ? 1 : 0;
            if (__syncthreads_or(dead)) break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    if (!ok_all) {
        if (tid == 0) { // the solve is over for this rank: every later launch of the block exits at once, the host sees
            P.st->exchange_timeout = 1; // `done`, finds the flag and lets the ranks agree on the all-reduce path
            P.st->done = 1;
        }
        if (tid < R) __hip_atomic_store((gu32 *)P.inbox[tid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    // (4) block 0: totals in rank order into slot 0, the other slots cleared
    if (blockIdx.x == 0) {
        if (tid < 4) {
            const double *rec = (const double *)s_recs;
            double t = 0.0;
            for (int r = 0; r < R; ++r) t += rec[4 * r + tid];
            P.buf[(size_t)tid * P.g_all] = t;
        }
        for (int i = 1 + tid; i < P.g_all; i += 256) {
#pragma unroll
            for (int c = 0; c < 4; ++c) P.buf[(size_t)c * P.g_all + i] = 0.0;
        }
    }
}

void stream_exchange_launch(double *buf, int32_t g_all, int32_t n_iface, int32_t rank, int32_t nranks, int32_t own0,
                            int32_t own1, int32_t fpar, uint32_t spin_limit, const int32_t *iface,
                            const uint8_t *iface_readers, void *const *inboxes, FusedState *st, hipStream_t s)
{
    StreamExchangeParams P = {};
    P.buf = buf;
    P.g_all = g_all;
    P.n_iface = n_iface;
    P.rank = rank;
    P.nranks = nranks;
    P.own0 = own0;
    P.own1 = own1;
    P.fpar = fpar;
    P.spin_limit = spin_limit;
    P.iface = iface;
    P.iface_readers = iface_readers;
    for (int r = 0; r < nranks && r < 8; ++r) P.inbox[r] = (uint8_t *)inboxes[r];
    P.st = st;
    int blocks = (n_iface + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 32 ? 32 : blocks);
    k_stream_exchange<<<blocks, 256, 0, s>>>(P);
}

int persist_threads()
{
    const char *e = getenv("MAG_TUNE_PERSIST_THREADS");
    const int t = e ? atoi(e) : kPersistThreadsDefault;
#ifdef MAG_PERSIST_768 // the 768 x 3 shape is only instantiated on request: measured 4 % slower (see the top of this file)
    return t == 768 ? 768 : 512;
#else
    (void)t;
    return 512;
#endif
}

int persist_tiles_per_wg(int32_t B, int threads)
{
    return B == 256 || B == 512 ? persist_npt(threads) * threads / B : 0; // whole tiles: 768 x 3 / 512 = 4
}

// dynamic LDS of a launch; every instantiation also carries 256 bytes of static LDS (the library's __syncthreads_and / _or),
// which the host's fit test adds (kPersistStaticLds)
size_t persist_lds_bytes(int32_t B, int32_t cap, int32_t maxh, int threads, int eb_mode, int32_t pool, bool mg)
{
    const size_t tiles = (size_t)persist_tiles_per_wg(B, threads);
    const size_t capx = eb_mode == 2 ? (size_t)B : (size_t)cap; // with overflow blocks the first area holds q of the owned nodes only
    return tiles * (capx + (size_t)cap + (size_t)maxh + (size_t)B) * 16 + 2 * 256 * 16 +
           (4 * ((size_t)threads / 64) + 4 + 4 * 32 + kPersistPartDoubles) * 8 + 16 + (eb_mode == 2 ? 32 * (size_t)pool : 0) +
           (mg ? 4 * (size_t)persist_npt(threads) * (size_t)threads : 0); // several ranks: the interface words (s_opk)
}

template <int THREADS>
static void persist_launch_t(const PersistParams &P, int32_t B, int32_t grid, size_t lds, int eb_mode, hipStream_t s)
{
    if (P.nranks > 1) {
        if (eb_mode == 2) {
            if (B == 256)
                k_cg_persist<256, true, THREADS, 2><<<grid, THREADS, lds, s>>>(P);
            else
                k_cg_persist<512, true, THREADS, 2><<<grid, THREADS, lds, s>>>(P);
        } else if (eb_mode == 1) {
            if (B == 256)
                k_cg_persist<256, true, THREADS, 1><<<grid, THREADS, lds, s>>>(P);
            else
                k_cg_persist<512, true, THREADS, 1><<<grid, THREADS, lds, s>>>(P);
        } else if (B == 256)
            k_cg_persist<256, true, THREADS, 0><<<grid, THREADS, lds, s>>>(P);
        else
            k_cg_persist<512, true, THREADS, 0><<<grid, THREADS, lds, s>>>(P);
    } else if (eb_mode == 2) {
        if (B == 256)
            k_cg_persist<256, false, THREADS, 2><<<grid, THREADS, lds, s>>>(P);
        else
            k_cg_persist<512, false, THREADS, 2><<<grid, THREADS, lds, s>>>(P);
    } else if (eb_mode == 1) {
        if (B == 256)
            k_cg_persist<256, false, THREADS, 1><<<grid, THREADS, lds, s>>>(P);
        else if constexpr (MAG_PERSIST_SPLIT_K4 && THREADS == 512)
            persist_launch_eb1_k4(P, grid, lds, s); // (persist_k4.o: the same instantiation under another scheduler)
        else
            k_cg_persist<512, false, THREADS, 1><<<grid, THREADS, lds, s>>>(P);
    } else if (B == 256)
        k_cg_persist<256, false, THREADS, 0><<<grid, THREADS, lds, s>>>(P);
    else
        k_cg_persist<512, false, THREADS, 0><<<grid, THREADS, lds, s>>>(P);
}

// eb_mode: 1 every row of the mesh qualifies for the edge-block instantiation (ring16's flag), 2 with overflow records in LDS
// (the host has checked the pool against the LDS), 0 the triangle walk
void persist_launch(const PersistParams &P, int32_t B, int32_t grid, int threads, int eb_mode, hipStream_t s)
{
    if (!kPersistEdgeBlocks) eb_mode = 0;
    const size_t lds = persist_lds_bytes(B, P.cap, P.maxh, threads, eb_mode, P.pool_cap, P.nranks > 1);
#ifdef MAG_PERSIST_768
    if (threads == 768) return persist_launch_t<768>(P, B, grid, lds, 0, s);
#endif
    // The whole mesh in one workgroup: the instantiation without an exchange -- for the edge-block instantiations, whose tiles read
    // every sibling's node through LDS.  (The triangle walk keeps halo COPIES of sibling nodes in tiles whose rows do not fit its
    // registers, and advances them with q fetched from the granules: it goes through the exchange even alone on the grid.)
    // Fewer than four tiles per workgroup (meshes below 769 tiles, 393k nodes): the instantiation with as many node slots per
    // lane (NPTX) -- 0.45 us per iteration less than four slots of which some are dead, at two and three tiles as well.
    if (P.nranks == 1 && B == 512 && eb_mode != 0 && (grid == 1 || (kPersistOneTile && P.tiles_per_wg < 4))) {
        const bool one = grid == 1;
        const int npt = kPersistOneTile && P.tiles_per_wg >= 1 && P.tiles_per_wg < 4 ? P.tiles_per_wg : 0;
#define MAG_PERSIST_CASE(EBM_, ONE_, NPTX_)                                                                                       \
    if (eb_mode == EBM_ && one == ONE_ && npt == NPTX_) k_cg_persist<512, false, 512, EBM_, ONE_, NPTX_><<<grid, 512, lds, s>>>(P)
#define MAG_PERSIST_CASES(EBM_, ONE_)                                                                                             \
    MAG_PERSIST_CASE(EBM_, ONE_, 1);                                                                                              \
    MAG_PERSIST_CASE(EBM_, ONE_, 2);                                                                                              \
    MAG_PERSIST_CASE(EBM_, ONE_, 3)
        MAG_PERSIST_CASES(1, false);
        MAG_PERSIST_CASES(2, false);
        MAG_PERSIST_CASES(1, true);
        MAG_PERSIST_CASES(2, true);
        MAG_PERSIST_CASE(1, true, 0);
        MAG_PERSIST_CASE(2, true, 0);
#undef MAG_PERSIST_CASES
#undef MAG_PERSIST_CASE
        return;
    }
    // ... and across ranks (a 1M-triangle mesh over four or eight GPUs is one tile per workgroup)
    if (kPersistOneTile && P.nranks > 1 && B == 512 && eb_mode != 0 && P.tiles_per_wg >= 1 && P.tiles_per_wg < 4) {
        const int npt = P.tiles_per_wg;
#define MAG_PERSIST_CASE(EBM_, NPTX_)                                                                                             \
    if (eb_mode == EBM_ && npt == NPTX_) k_cg_persist<512, true, 512, EBM_, false, NPTX_><<<grid, 512, lds, s>>>(P)
        MAG_PERSIST_CASE(1, 1);
        MAG_PERSIST_CASE(1, 2);
        MAG_PERSIST_CASE(1, 3);
        MAG_PERSIST_CASE(2, 1);
        MAG_PERSIST_CASE(2, 2);
        MAG_PERSIST_CASE(2, 3);
#undef MAG_PERSIST_CASE
        return;
    }
    persist_launch_t<512>(P, B, grid, lds, eb_mode, s);
}

int persist_block_entries() { return kPersistBlockEntries; }

// The edge blocks of every node (cg_device.h, ring_blocks), once per solve: one thread per node of the padded Hilbert order
// reads its ring words (tile-local ids) and the coordinates of its neighbours from the tile tables in memory -- owned nodes
// from xyP, halo nodes from the tile's contiguous halo copy -- and writes the 3 NB numbers, value-major (the on-chip kernel's
// loads are coalesced).  Only launched for meshes k_ring16 found to qualify (every row one fan of at most NB entries, or
// NB + 1 closing onto the first).
template <int B>
__global__ void __launch_bounds__(256) k_edge_blocks(const PersistParams P, double *kbg)
{
    constexpr int NB = kPersistBlockEntries, NW = (NB + 2) / 2;
    // (several ranks: the blocks of this rank's own tiles only -- the launcher's grid covers [t0, t1))
    const int64_t nd = (P.nranks > 1 ? (int64_t)P.t0 * B : 0) + (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int32_t t = (int32_t)(nd / B), lt = (int32_t)(nd % B);
    if (t >= (P.nranks > 1 ? P.t1 : P.T)) return;
    const TileMeta tm = P.meta[t];
    double kb[3 * NB];
#pragma unroll
    for (int c = 0; c < 3 * NB; ++c) kb[c] = 0.0;
    if (nd < P.N && tm.ent > 0) {
        uint32_t w[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) w[k] = k < tm.deg ? P.ell16[tm.ell_off + lt + (int64_t)k * B] : 0xffffffffu;
        auto xy_of = [&](uint32_t lid) -> double2 {
            if (lid < (uint32_t)B) {
                const int64_t g = (int64_t)t * B + lid;
                return g < P.N ? P.xyP[g] : make_double2(0.0, 0.0);
            }
            return P.halo_xy[tm.hoff + (int32_t)(lid - B)];
        };
        ring_blocks<NW, NB, 0xfffu>(w, tm.ent < 2 * NW ? tm.ent : 2 * NW, xy_of, P.xyP[nd], P.c0, P.nu, P.h, kb);
    }
#pragma unroll
    for (int c = 0; c < 3 * NB; ++c) kbg[(int64_t)c * P.kb_stride + nd] = kb[c];
}

// The same for meshes whose rows are single fans of ANY length (EBM == 2).  The row is streamed from the ring table, one
// triangle at a time: triangle k (between entries k - 1 and k) contributes its K_ab to block k - 1 and its K_ac to block k --
// or to block 0 when it closes the fan (the row's last entry repeats its first: row_info bit 6), so a closed fan of valence
// v is v blocks.  Block j >= 1 is complete once triangle j + 1 is through; block 0 is held to the end.  Blocks below NB go to
// the value-major arrays the registers are loaded from, block j >= NB to the node's overflow records, at ovf_off[node] +
// j - NB, with the ring entry (tile-local id) it multiplies.  Per block the same two-term sums in the same order as
// ring_blocks: a structured mesh gets the same bits from either kernel.
template <int B>
__global__ void __launch_bounds__(256) k_edge_blocks_ovf(const PersistParams P, double *kbg)
{
    constexpr int NB = kPersistBlockEntries;
    const int64_t nd = (P.nranks > 1 ? (int64_t)P.t0 * B : 0) + (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int32_t t = (int32_t)(nd / B), lt = (int32_t)(nd % B);
    if (t >= (P.nranks > 1 ? P.t1 : P.T)) return;
    const TileMeta tm = P.meta[t];
    const uint32_t info = nd < P.N ? P.row_info[nd] : 0u;
    const int32_t n = (info & 0x80u) ? (int32_t)(info & 63u) : 0;
    const bool closed = (info & 0x40u) != 0;
    const int32_t nblk = closed ? n - 1 : n;
    auto emit = [&](int32_t j, double v0, double v1, double v2, uint32_t id) {
        if (j < NB) {
            kbg[(int64_t)(3 * j + 0) * P.kb_stride + nd] = v0;
            kbg[(int64_t)(3 * j + 1) * P.kb_stride + nd] = v1;
            kbg[(int64_t)(3 * j + 2) * P.kb_stride + nd] = v2;
        } else {
            double2 *rec = (double2 *)P.ovf_rec + 2 * ((int64_t)P.ovf_off[nd] + (j - NB));
            rec[0] = make_double2(v0, v1);
            rec[1] = make_double2(v2, __hiloint2double(0, (int)id));
        }
    };
    for (int32_t j = nblk > 0 ? nblk : 0; j < NB; ++j) emit(j, 0.0, 0.0, 0.0, 0u); // blocks the row does not have
    // An open fan of more than NB entries keeps its LAST block in register position NB - 1 and its middle blocks NB - 1 ..
    // n - 2 in the pool: u_first and u_last, which the walk's telescoped antisymmetric part needs, are then always register
    // entries (the on-chip kernel moves the row's last ring entry into position NB - 1 to match).
    const bool open_long = !closed && n > NB;
    auto place = [&](int32_t j, double v0, double v1, double v2, uint32_t id) {
        if (open_long && j >= NB - 1) j = j == n - 1 ? NB - 1 : j + 1;
        emit(j, v0, v1, v2, id);
    };
    if (n < 2) {
        if (nblk == 1) emit(0, 0.0, 0.0, 0.0, 0u); // one entry, no triangle
        return;
    }
    auto entry = [&](int32_t k) {
        const uint32_t ww = P.ell16[tm.ell_off + lt + (int64_t)(k >> 1) * B];
        return (k & 1) ? (ww >> 16) : (ww & 0xffffu);
    };
    auto xy_of = [&](uint32_t lid) -> double2 {
        if (lid < (uint32_t)B) {
            const int64_t g = (int64_t)t * B + lid;
            return g < P.N ? P.xyP[g] : make_double2(0.0, 0.0);
        }
        return P.halo_xy[tm.hoff + (int32_t)(lid - B)];
    };
    const double2 ca = P.xyP[nd];
    const double2 z = make_double2(0.0, 0.0), ex = make_double2(1.0, 0.0), ey = make_double2(0.0, 1.0);
    const uint32_t id0 = entry(0) & 0xfffu;
    double2 pd;
    {
        const double2 cxy = xy_of(id0);
        pd = make_double2(cxy.x - ca.x, cxy.y - ca.y);
    }
    double k0[3] = {0.0, 0.0, 0.0};    // block 0, held to the end
    double carry[3] = {0.0, 0.0, 0.0}; // K_ac of the triangle before entry k - 1: the first term of block k - 1
    uint32_t idprev = id0;
    for (int32_t k = 1; k < n; ++k) {
        const uint32_t id = entry(k) & 0xfffu;
        const double2 cxy = xy_of(id);
        const double2 d = make_double2(cxy.x - ca.x, cxy.y - ca.y);
        const double twoA = fma(pd.x, d.y, -(d.x * pd.y));
        const double wt = P.c0 * fast_rcp(twoA);
        double b00 = 0.0, b10 = 0.0, b01 = 0.0, b11 = 0.0, c00 = 0.0, c10 = 0.0, c01 = 0.0, c11 = 0.0;
        fan_force_w<double2, double>(pd, ex, d, z, wt, P.nu, P.h, b00, b10); // K_ab, first column
        fan_force_w<double2, double>(pd, ey, d, z, wt, P.nu, P.h, b01, b11);
        fan_force_w<double2, double>(pd, z, d, ex, wt, P.nu, P.h, c00, c10); // K_ac
        fan_force_w<double2, double>(pd, z, d, ey, wt, P.nu, P.h, c01, c11);
        const double bs = 0.5 * (b01 + b10), cs = 0.5 * (c01 + c10);
        if (k == 1) { // block 0 = K_ab of triangle 1 (+ K_ac of the closing triangle, below)
            k0[0] = 0.0 + b00;
            k0[1] = 0.0 + bs;
            k0[2] = 0.0 + b11;
        } else {
            place(k - 1, carry[0] + b00, carry[1] + bs, carry[2] + b11, idprev);
        }
        carry[0] = 0.0 + c00;
        carry[1] = 0.0 + cs;
        carry[2] = 0.0 + c11;
        pd = d;
        idprev = id;
    }
    if (closed) { // the last triangle closes onto entry 0
        place(0, k0[0] + carry[0], k0[1] + carry[1], k0[2] + carry[2], id0);
    } else {
        place(0, k0[0], k0[1], k0[2], id0);
        place(n - 1, carry[0], carry[1], carry[2], idprev);
    }
}

void edge_blocks_build(const PersistParams &P, int32_t B, double *kblocks, int eb_mode, hipStream_t s)
{
    const int64_t npad = (int64_t)(P.nranks > 1 ? P.t1 - P.t0 : P.T) * B; // several ranks: this rank's own tiles
    const unsigned blocks = (unsigned)((npad + 255) / 256);
    if (eb_mode == 2) {
        if (B == 256)
            k_edge_blocks_ovf<256><<<blocks, 256, 0, s>>>(P, kblocks);
        else
            k_edge_blocks_ovf<512><<<blocks, 256, 0, s>>>(P, kblocks);
    } else if (B == 256)
        k_edge_blocks<256><<<blocks, 256, 0, s>>>(P, kblocks);
    else
        k_edge_blocks<512><<<blocks, 256, 0, s>>>(P, kblocks);
}

int persist_stamp_words() { return kStampPhases + 1; }
bool persist_stamps_built()
{
#ifdef MAG_PERSIST_STAMPS
    return true;
#else
    return false;
#endif
}

// bit 3 of the node mask, for the on-chip kernel with `k` tiles per workgroup: some tile of THIS rank's range [t0, t1) reads
// the node THROUGH MEMORY -- a tile of another workgroup, or a sibling tile whose rows do not all fit the registers (it
// keeps its halo copies).  One block per reading tile.  (Tiles of other ranks read through the inboxes, not through this
// GPU's granules: they mark nothing here.)
__global__ void __launch_bounds__(256) k_mark_external(const int32_t *halo_g, const TileMeta *meta, int32_t B, int32_t k,
                                                       int32_t max_reg_entries, int32_t t0, int32_t t1, uint8_t *maskP)
{
    const int32_t t = t0 + (int32_t)blockIdx.x;
    const TileMeta tm = meta[t];
    for (int32_t h = threadIdx.x; h < tm.nh; h += 256) {
        const int64_t g = halo_g[tm.hoff + h];
        const int32_t ot = (int32_t)(g / B);
        const bool sibling = ot >= t0 && ot < t1 && (ot - t0) / k == (t - t0) / k && tm.ent <= max_reg_entries;
        if (!sibling) atomicOr((unsigned int *)(maskP + (g & ~(int64_t)3)), 8u << (8 * (g & 3)));
    }
}

void mark_external(const int32_t *halo_g, const TileMeta *meta, int32_t t0, int32_t t1, int32_t B, int32_t k, uint8_t *maskP,
                   bool all_rows, hipStream_t s)
{
    if (t1 > t0)
        k_mark_external<<<t1 - t0, 256, 0, s>>>(halo_g, meta, B, k, all_rows ? 0x7fffffff : 2 * kPersistRegs, t0, t1, maskP);
}

// bit 2 of the node mask: some tile reads this node through its halo list, so its owner must publish q
__global__ void __launch_bounds__(256) k_mark_published(const int32_t *halo_g, int64_t halo_total, uint8_t *maskP)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= halo_total) return;
    // byte-wide atomic OR does not exist; neighbouring bytes belong to other nodes, so OR the containing word
    const int64_t g = halo_g[i];
    atomicOr((unsigned int *)(maskP + (g & ~(int64_t)3)), 4u << (8 * (g & 3)));
}

void mark_published(const int32_t *halo_g, int64_t halo_total, uint8_t *maskP, hipStream_t s)
{
    if (halo_total > 0) k_mark_published<<<(unsigned)((halo_total + 255) / 256), 256, 0, s>>>(halo_g, halo_total, maskP);
}

} // namespace magk
#endif // MAG_PERSIST_TU_K4
