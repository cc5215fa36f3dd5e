// Internal: sum-all-reduce transport for the multi-GPU CG (one process per GPU).
//   * RCCL over xGMI, resolved with dlopen at first use so a single-GPU caller
//     never loads librccl (and a host process that already loaded RCCL -- e.g.
//     bench.py through torch -- shares that copy: same SONAME librccl.so.1).
//   * host callback, for tests: D2H -> caller's reduction (gloo) -> H2D.
#pragma once
#include <cstdint>
#include <string>

#include <hip/hip_runtime.h>

#include "magnetite_hip.h"

namespace magc {

int get_unique_id(void *id_out);

struct Comm {
    int nranks = 1, rank = 0;
    void *nccl = nullptr;
    mag_allreduce_fn cb = nullptr;
    void *cb_user = nullptr;
    double *h_stage = nullptr; // pinned staging for the callback transport
    size_t h_cap = 0;

    bool distributed() const { return nranks > 1; }
    int init_rccl(const void *unique_id, int nranks_, int rank_, hipStream_t s, std::string &msg);
    int init_callback(int nranks_, int rank_, mag_allreduce_fn fn, void *user, std::string &msg);
    // in-place sum over ranks of count doubles at dev_buf, ordered on stream s
    int allreduce_sum(double *dev_buf, int64_t count, hipStream_t s, std::string &msg);
    // every rank contributes `count` doubles at dev_send; dev_recv receives nranks * count doubles in rank order
    // (RCCL: ncclAllGather; callback transport: placed into a zeroed buffer and sum-all-reduced -- same result)
    int allgather(const double *dev_send, double *dev_recv, int64_t count, hipStream_t s, std::string &msg);
    // true when allreduce_sum only enqueues work on s (RCCL); false when it synchronises (callback)
    bool stream_ordered() const { return nccl != nullptr; }
    // ranks of the RCCL communicator as RCCL itself reports them (ncclCommCount); 0 without one, -1 on error
    int rccl_count() const;
    void destroy();
};

} // namespace magc
