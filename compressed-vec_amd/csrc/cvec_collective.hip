// cvec_collective.hip - the one exchange of the path across GPUs: the all-reduce of a result block.
//
// Vectors are independent, so every rank filters its own shard and nothing of the data path crosses GPUs; what the
// ranks have to agree on is the match count (reference: filter::count_hits returns one number for one vector,
// src/filter.rs:296-299; over a sharded batch the number is the sum over the shards).  The library does not link
// RCCL: the entry points below bind librccl at their first use (dlopen), so a process that never calls them needs no
// RCCL, and a process that already has RCCL loaded (its own communicator) keeps using that copy.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "cvec_device.hpp"

namespace cvec {
cvec_status record_hip_error(hipError_t e, const char *what); // cvec_api.hip
void record_message(const char *msg);                         // cvec_api.hip
} // namespace cvec

namespace {

// the slice of RCCL's C API that is used (rccl.h: ncclResult_t is an int, ncclSuccess = 0)
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_group)(void);
typedef int (*fn_unique_id)(void *);
typedef int (*fn_comm_init_rank)(void **, int, cvec_comm_id, int);
typedef int (*fn_comm_destroy)(void *);
typedef const char *(*fn_error_string)(int);
constexpr int NCCL_UINT64 = 5; // ncclUint64
constexpr int NCCL_SUM = 0, NCCL_MAX = 2;

struct Rccl {
    void *handle = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
    fn_unique_id unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_error_string error_string = nullptr;
    char why[256] = "";
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void bind_rccl() {
    Rccl &r = g_rccl;
    // a copy that the process has loaded already (the caller's communicator belongs to it), else the ROCm installation's
    const char *env = getenv("CVEC_RCCL_PATH");
    const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : {"librccl.so.1", "librccl.so"}) {
        if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    }
    for (const char *n : names) {
        if (!r.handle && n && *n) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.handle) {
        snprintf(r.why, sizeof(r.why), "librccl not found (%s); set CVEC_RCCL_PATH", dlerror());
        return;
    }
    r.all_reduce = (fn_all_reduce)dlsym(r.handle, "ncclAllReduce");
    r.group_start = (fn_group)dlsym(r.handle, "ncclGroupStart");
    r.group_end = (fn_group)dlsym(r.handle, "ncclGroupEnd");
    r.unique_id = (fn_unique_id)dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(r.handle, "ncclCommInitRank");
    r.comm_destroy = (fn_comm_destroy)dlsym(r.handle, "ncclCommDestroy");
    r.error_string = (fn_error_string)dlsym(r.handle, "ncclGetErrorString");
    if (!r.all_reduce || !r.group_start || !r.group_end || !r.unique_id || !r.comm_init_rank || !r.comm_destroy) {
        snprintf(r.why, sizeof(r.why), "librccl lacks one of ncclAllReduce / ncclGroupStart / ncclGroupEnd / ncclGetUniqueId / "
                                       "ncclCommInitRank / ncclCommDestroy");
        return;
    }
    r.ok = true;
}

cvec_status rccl_ready() {
    std::call_once(g_rccl_once, bind_rccl);
    if (!g_rccl.ok) {
        cvec::record_message(g_rccl.why);
        return CVEC_ERR_HIP;
    }
    return CVEC_OK;
}

cvec_status rccl_fail(int rc, const char *what) {
    char msg[256];
    snprintf(msg, sizeof(msg), "%s: %s (ncclResult %d)", what, g_rccl.error_string ? g_rccl.error_string(rc) : "RCCL error", rc);
    cvec::record_message(msg);
    return CVEC_ERR_HIP;
}

} // namespace

extern "C" {

cvec_status cvec_comm_unique_id(cvec_comm_id *id) {
    if (!id) return CVEC_ERR_BAD_ARGUMENT;
    cvec_status st = rccl_ready();
    if (st) return st;
    static_assert(sizeof(cvec_comm_id) == 128, "ncclUniqueId is 128 bytes");
    const int rc = g_rccl.unique_id(id);
    return rc ? rccl_fail(rc, "ncclGetUniqueId") : CVEC_OK;
}

cvec_status cvec_comm_init_rank(const cvec_comm_id *id, int32_t n_ranks, int32_t rank, void **comm) {
    if (!id || !comm || n_ranks < 1 || rank < 0 || rank >= n_ranks) return CVEC_ERR_BAD_ARGUMENT;
    cvec_status st = rccl_ready();
    if (st) return st;
    void *c = nullptr;
    const int rc = g_rccl.comm_init_rank(&c, n_ranks, *id, rank);
    if (rc) return rccl_fail(rc, "ncclCommInitRank");
    *comm = c;
    return CVEC_OK;
}

cvec_status cvec_comm_destroy(void *comm) {
    if (!comm) return CVEC_OK;
    cvec_status st = rccl_ready();
    if (st) return st;
    const int rc = g_rccl.comm_destroy(comm);
    return rc ? rccl_fail(rc, "ncclCommDestroy") : CVEC_OK;
}

cvec_status cvec_allreduce_counts(uint64_t *d_result, uint32_t n_blocks, void *comm, void *stream) {
    if (!d_result || !comm || n_blocks == 0) return CVEC_ERR_BAD_ARGUMENT;
    cvec_status st = rccl_ready();
    if (st) return st;
    hipStream_t s = (hipStream_t)stream;
    // word 0 (total) and word 1 (failed vectors) add up over the ranks, word 2 (largest status) takes the maximum,
    // word 3 (the encoder's section count) adds up: three small collectives per block, all of them in one group, so
    // that a bucket of blocks costs one launch
    int rc = g_rccl.group_start();
    if (rc) return rccl_fail(rc, "ncclGroupStart");
    for (uint32_t b = 0; b < n_blocks && !rc; b++) {
        uint64_t *r = d_result + (size_t)b * CVEC_RESULT_WORDS;
        rc = g_rccl.all_reduce(r, r, 2, NCCL_UINT64, NCCL_SUM, comm, s);
        if (!rc) rc = g_rccl.all_reduce(r + 2, r + 2, 1, NCCL_UINT64, NCCL_MAX, comm, s);
        if (!rc) rc = g_rccl.all_reduce(r + 3, r + 3, 1, NCCL_UINT64, NCCL_SUM, comm, s);
    }
    const int rc2 = g_rccl.group_end();
    if (rc) return rccl_fail(rc, "ncclAllReduce");
    if (rc2) return rccl_fail(rc2, "ncclGroupEnd");
    return CVEC_OK;
}

} // extern "C"
