// RCCL / callback transports behind magc::Comm (see comm.h).
#include "comm.h"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include <rccl/rccl.h>

namespace magc {

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.err = std::string("dlopen(librccl.so.1) failed: ") + dlerror();
            return;
        }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
        r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
        r.CommCount = (decltype(r.CommCount))dlsym(r.handle, "ncclCommCount");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.AllGather || !r.CommDestroy || !r.GetErrorString)
            r.err = "librccl is missing a required symbol";
    });
    return r;
}

} // namespace

int get_unique_id(void *id_out)
{
    static_assert(sizeof(ncclUniqueId) == MAG_UNIQUE_ID_BYTES, "unique id size");
    if (!id_out) return MAG_ERR_BAD_ARGS;
    Rccl &r = rccl();
    if (!r.err.empty()) return MAG_ERR_RCCL;
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return MAG_ERR_RCCL;
    memcpy(id_out, &id, sizeof id);
    return MAG_OK;
}

int Comm::init_rccl(const void *unique_id, int nranks_, int rank_, hipStream_t, std::string &msg)
{
    if (!unique_id || nranks_ < 1 || rank_ < 0 || rank_ >= nranks_) {
        msg = "bad communicator arguments";
        return MAG_ERR_BAD_ARGS;
    }
    Rccl &r = rccl();
    if (!r.err.empty()) {
        msg = r.err;
        return MAG_ERR_RCCL;
    }
    destroy();
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclComm_t c = nullptr;
    const ncclResult_t rc = r.CommInitRank(&c, nranks_, id, rank_);
    if (rc != ncclSuccess) {
        msg = std::string("ncclCommInitRank failed: ") + r.GetErrorString(rc);
        return MAG_ERR_RCCL;
    }
    nccl = (void *)c;
    nranks = nranks_;
    rank = rank_;
    return MAG_OK;
}

int Comm::init_callback(int nranks_, int rank_, mag_allreduce_fn fn, void *user, std::string &msg)
{
    if (!fn || nranks_ < 1 || rank_ < 0 || rank_ >= nranks_) {
        msg = "bad communicator arguments";
        return MAG_ERR_BAD_ARGS;
    }
    destroy();
    cb = fn;
    cb_user = user;
    nranks = nranks_;
    rank = rank_;
    return MAG_OK;
}

int Comm::allreduce_sum(double *dev_buf, int64_t count, hipStream_t s, std::string &msg)
{
    if (count <= 0) return MAG_OK;
    if (nccl) {
        Rccl &r = rccl();
        const ncclResult_t rc = r.AllReduce(dev_buf, dev_buf, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)nccl, s);
        if (rc != ncclSuccess) {
            msg = std::string("ncclAllReduce failed: ") + r.GetErrorString(rc);
            return MAG_ERR_RCCL;
        }
        return MAG_OK;
    }
    if (cb) {
        const size_t bytes = 8 * (size_t)count;
        if (bytes > h_cap) {
            if (h_stage) (void)hipHostFree(h_stage);
            h_stage = nullptr;
            h_cap = 0;
            if (hipHostMalloc((void **)&h_stage, bytes * 2, hipHostMallocDefault) != hipSuccess) {
                msg = "hipHostMalloc failed for the all-reduce staging buffer";
                return MAG_ERR_HIP;
            }
            h_cap = bytes * 2;
        }
        if (hipMemcpyAsync(h_stage, dev_buf, bytes, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            msg = "D2H staging of the all-reduce buffer failed";
            return MAG_ERR_HIP;
        }
        if (cb(cb_user, h_stage, count) != 0) {
            msg = "all-reduce callback reported failure";
            return MAG_ERR_RCCL;
        }
        if (hipMemcpyAsync(dev_buf, h_stage, bytes, hipMemcpyHostToDevice, s) != hipSuccess) {
            msg = "H2D staging of the all-reduce buffer failed";
            return MAG_ERR_HIP;
        }
        return MAG_OK;
    }
    if (nranks > 1) {
        msg = "communicator not initialised";
        return MAG_ERR_STATE;
    }
    return MAG_OK;
}

int Comm::allgather(const double *dev_send, double *dev_recv, int64_t count, hipStream_t s, std::string &msg)
{
    if (count <= 0) return MAG_OK;
    if (nccl) {
        Rccl &r = rccl();
        const ncclResult_t rc = r.AllGather(dev_send, dev_recv, (size_t)count, ncclDouble, (ncclComm_t)nccl, s);
        if (rc != ncclSuccess) {
            msg = std::string("ncclAllGather failed: ") + r.GetErrorString(rc);
            return MAG_ERR_RCCL;
        }
        return MAG_OK;
    }
    // callback transport / no transport: own segment into a zeroed buffer, summed over ranks
    if (hipMemsetAsync(dev_recv, 0, 8 * (size_t)count * (size_t)nranks, s) != hipSuccess ||
        hipMemcpyAsync(dev_recv + (size_t)rank * (size_t)count, dev_send, 8 * (size_t)count, hipMemcpyDeviceToDevice, s) !=
            hipSuccess) {
        msg = "staging of the all-gather buffer failed";
        return MAG_ERR_HIP;
    }
    return allreduce_sum(dev_recv, count * nranks, s, msg);
}

int Comm::rccl_count() const
{
    if (!nccl) return 0;
    Rccl &r = rccl();
    int n = 0;
    if (!r.CommCount || r.CommCount((ncclComm_t)nccl, &n) != ncclSuccess) return -1;
    return n;
}

void Comm::destroy()
{
    if (nccl) {
        Rccl &r = rccl();
        if (r.CommDestroy) (void)r.CommDestroy((ncclComm_t)nccl);
        nccl = nullptr;
    }
    if (h_stage) {
        (void)hipHostFree(h_stage);
        h_stage = nullptr;
        h_cap = 0;
    }
    cb = nullptr;
    cb_user = nullptr;
    nranks = 1;
    rank = 0;
}

} // namespace magc
