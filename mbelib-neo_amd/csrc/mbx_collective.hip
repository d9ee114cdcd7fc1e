// mbx_collective.hip -- the one collective of the path: RCCL broadcast of the constant-table blob at start-up, for hosts
// that are not Python (north_star: "RCCL broadcast of the shared codebook tables over xGMI only").
// bench.py does the same through torch.distributed (mbelib-neo_amd/parallel.py); a C host gets it from here.
//
// Frames are embarrassingly parallel, streams shard across ranks with no data-path collective, so this is ALL the
// communication there is: rank `root` holds the blob (71,908 bytes), one ncclBroadcast hands it to every rank's GPU over
// xGMI, every rank uploads its copy with mbx_init() and an ncclAllReduce (min, max) of the per-rank table checksums
// proves that all ranks decode with the same tables.  The reference has no counterpart (single-process CPU library).
//
// RCCL is bound at RUN TIME (dlopen "librccl.so.1"): libmbx_hip.so carries no link-time dependency on it, so a
// single-GPU host that never calls these entry points needs no RCCL at all, and inside a process that already has one
// loaded (PyTorch) the same instance is used.  MBX_RCCL_LIBRARY in the environment names another library to bind instead
// (read once, at the first call): a site's own RCCL build -- and tests/fake_rccl.c, whose collectives rendezvous host
// threads on ONE device, which is how the N > 1 control flow below (non-root ranks, mismatching tables, a rank that fails)
// is exercised on a single-GPU box.  Host code only.
//
// What "no rank hangs" covers: from the first collective on, a rank that fails LOCALLY -- its copy of the blob does not
// arrive or does not pass mbx_init's magic / checksum test, a copy or a collective call returns an error -- still makes every
// remaining collective call of the sequence (broadcast, all-reduce min, all-reduce max) and contributes a pair that cannot
// agree, so its peers return MBX_EBADTABLE instead of waiting for it.  What it cannot cover: a rank that never reaches the
// first collective (its device is gone, the 72 KB staging allocation fails) -- the peers of a rank that is not there wait in
// ncclBroadcast like the peers of any dead RCCL rank; that is the launcher's time-out to catch.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and enumerators only

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "mbx.h"

void mbx_set_error_text(const char* text);   // mbx_api.hip

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

int cfail(int code, const char* what, const char* detail = nullptr) {
    char buf[256];
    snprintf(buf, sizeof(buf), "collective: %s%s%s", what, detail ? ": " : "", detail ? detail : "");
    mbx_set_error_text(buf);
    return code;
}

const Rccl* rccl() {   // nullptr (with the error text set) when RCCL cannot be loaded
    static Rccl r;
    static std::once_flag once;
    static bool ok = false;
    std::call_once(once, [] {
        const char* override_name = getenv("MBX_RCCL_LIBRARY");
        if (override_name && override_name[0]) {   // exactly that library or none: a typo must not fall back silently
            r.handle = dlopen(override_name, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (r.handle) {
                    break;
                }
            }
        }
        if (!r.handle) {
            return;
        }
        auto sym = [&](const char* n) { return dlsym(r.handle, n); };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommCount && r.CommUserRank && r.Broadcast && r.AllReduce
             && r.GetErrorString;
    });
    if (!ok) {
        (void)cfail(MBX_ENODEVICE, "RCCL (librccl.so.1) could not be loaded", dlerror());
        return nullptr;
    }
    return &r;
}

#define N_TRY(R, expr)                                                   \
    do {                                                                 \
        ncclResult_t n_ = (expr);                                        \
        if (n_ != ncclSuccess) {                                         \
            return cfail(MBX_ENODEVICE, #expr, (R)->GetErrorString(n_)); \
        }                                                                \
    } while (0)
#define H_TRY(expr)                                                      \
    do {                                                                 \
        hipError_t e_ = (expr);                                          \
        if (e_ != hipSuccess) {                                          \
            return cfail(MBX_ENODEVICE, #expr, hipGetErrorString(e_));   \
        }                                                                \
    } while (0)

struct DevBytes {
    void* p = nullptr;
    ~DevBytes() {
        if (p) {
            (void)hipFree(p);
        }
    }
};

// `d`: four words of device memory whose first two words hold the POISON pair (0, 0xffffffff) -- no set of honest values can
// reduce to equal words with it in -- written at allocation time (poison_words), so that a rank whose copy of its real pair
// fails still takes part with the poison.
int poison_words(uint32_t* d) {
    if (hipMemset(d, 0x00, sizeof(uint32_t)) != hipSuccess || hipMemset(d + 1, 0xff, sizeof(uint32_t)) != hipSuccess) {
        (void)hipGetLastError();
        return cfail(MBX_ENODEVICE, "collective: could not initialise the agreement words");
    }
    return 0;
}

// min and max over the ranks of one 32-bit value (two all-reduces of one word each); `failed`: this rank contributes the
// poison pair instead.  BOTH all-reduces are always issued, whatever fails locally on the way: the peers are in them.
// 0 when every rank passed the same value; MBX_EBADTABLE when they differ or a rank failed; MBX_ENODEVICE for a local error.
int agree(const Rccl* R, ncclComm_t c, uint32_t value, bool failed, uint32_t* d /* see poison_words */, uint32_t out[2], hipStream_t st) {
    int local_rc = 0;
    if (!failed) {
        const uint32_t in[2] = {value, value};
        if (hipMemcpyAsync(d, in, sizeof(in), hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            (void)hipGetLastError();
            local_rc = cfail(MBX_ENODEVICE, "agree: the copy of this rank's value failed (it takes part with the poison pair)");
            failed = true;
        }
    }
    const ncclResult_t n0 = R->AllReduce(d, d + 2, 1, ncclUint32, ncclMin, c, st);
    const ncclResult_t n1 = R->AllReduce(d + 1, d + 3, 1, ncclUint32, ncclMax, c, st);
    if (n0 != ncclSuccess || n1 != ncclSuccess) {
        local_rc = cfail(MBX_ENODEVICE, "agree: ncclAllReduce", R->GetErrorString(n0 != ncclSuccess ? n0 : n1));
    }
    out[0] = 0u;
    out[1] = 0xffffffffu;
    if (hipMemcpyAsync(out, d + 2, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        (void)hipGetLastError();
        out[0] = 0u;
        out[1] = 0xffffffffu;
        local_rc = cfail(MBX_ENODEVICE, "agree: the result did not come back on this rank");
    }
    if (local_rc < 0) {
        return local_rc;
    }
    return (out[0] == out[1] && !failed) ? 0 : MBX_EBADTABLE;
}

}  // namespace

extern "C" {

int mbx_comm_unique_id(void* id128) {
    const Rccl* R = rccl();
    if (!R) {
        return MBX_ENODEVICE;
    }
    if (!id128) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    ncclUniqueId id;
    N_TRY(R, R->GetUniqueId(&id));
    static_assert(sizeof(id) == MBX_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return 0;
}

int mbx_comm_init(void** comm, int nranks, const void* id128, int rank, int device) {
    const Rccl* R = rccl();
    if (!R) {
        return MBX_ENODEVICE;
    }
    if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    H_TRY(hipSetDevice(device));   // the communicator binds to the calling thread's current device
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    N_TRY(R, R->CommInitRank(&c, nranks, id, rank));
    *comm = c;
    return 0;
}

int mbx_comm_destroy(void* comm) {
    const Rccl* R = rccl();
    if (!R) {
        return MBX_ENODEVICE;
    }
    if (comm) {
        N_TRY(R, R->CommDestroy(static_cast<ncclComm_t>(comm)));
    }
    return 0;
}

int mbx_init_broadcast(void* comm, int root, int device, void* table_blob, size_t table_bytes, uint32_t* checksums_min_max, void* stream) {
    const Rccl* R = rccl();
    if (!R) {
        return MBX_ENODEVICE;
    }
    if (!comm || !table_blob || table_bytes != sizeof(mbx_tables)) {
        return cfail(MBE_STATUS_INVALID_ARGUMENT, "mbx_init_broadcast: communicator, a blob buffer of sizeof(mbx_tables) bytes on every rank");
    }
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    int nranks = 0, rank = -1;
    N_TRY(R, R->CommCount(c, &nranks));
    N_TRY(R, R->CommUserRank(c, &rank));
    if (root < 0 || root >= nranks) {
        return cfail(MBE_STATUS_INVALID_ARGUMENT, "mbx_init_broadcast: root is not a rank of the communicator");
    }
    // (failures up to here and in the three lines below happen BEFORE this rank's first collective call: see the header comment)
    H_TRY(hipSetDevice(device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    DevBytes blob, sums;
    H_TRY(hipMalloc(&blob.p, table_bytes));
    H_TRY(hipMalloc(&sums.p, 4 * sizeof(uint32_t)));
    {
        const int prc = poison_words(static_cast<uint32_t*>(sums.p));
        if (prc < 0) {
            return prc;
        }
    }
    // From here on every rank makes EVERY collective call of the sequence whatever happens to it locally: a rank that returned
    // early (a blob that fails mbx_init's magic / checksum test is exactly the case this function exists to catch) would leave
    // the others blocked in a collective for ever.  A failed rank contributes a pair that cannot agree and reports its own error.
    int local_rc = 0;
    auto local_fail = [&](const char* what, const char* detail) {
        if (local_rc == 0) {   // the first error is the one reported
            local_rc = cfail(MBX_ENODEVICE, what, detail);
        }
    };
    if (rank == root) {
        const hipError_t e = hipMemcpyAsync(blob.p, table_blob, table_bytes, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            local_fail("mbx_init_broadcast: the root's copy of the blob to its device failed", hipGetErrorString(e));
        }
    }
    {   // the one collective of the path: 71,908 bytes, root's GPU -> every GPU
        const ncclResult_t n = R->Broadcast(blob.p, blob.p, table_bytes, ncclUint8, root, c, st);
        if (n != ncclSuccess) {
            local_fail("mbx_init_broadcast: ncclBroadcast", R->GetErrorString(n));
        }
    }
    if (rank != root && local_rc == 0) {
        const hipError_t e = hipMemcpyAsync(table_blob, blob.p, table_bytes, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            local_fail("mbx_init_broadcast: the copy of the received blob to the host failed", hipGetErrorString(e));
        }
    }
    if (hipStreamSynchronize(st) != hipSuccess) {
        (void)hipGetLastError();
        local_fail("mbx_init_broadcast: the broadcast did not complete on this rank", nullptr);
    }
    if (local_rc == 0) {
        local_rc = mbx_init(device, table_blob, table_bytes);   // validates magic / version / checksum of what arrived
    }
    uint32_t out[2] = {0u, 0u};
    const int arc = agree(R, c, local_rc == 0 ? mbx_table_checksum() : 0u, local_rc != 0, static_cast<uint32_t*>(sums.p), out, st);
    if (checksums_min_max) {
        checksums_min_max[0] = out[0];
        checksums_min_max[1] = out[1];
    }
    if (local_rc < 0) {
        return local_rc;
    }
    if (arc == MBX_EBADTABLE) {
        return cfail(MBX_EBADTABLE, "mbx_init_broadcast: the table checksums differ between ranks (or a rank failed to initialise)");
    }
    return arc;
}

int mbx_comm_agree(void* comm, uint32_t value, uint32_t* min_max, void* stream) {
    const Rccl* R = rccl();
    if (!R) {
        return MBX_ENODEVICE;
    }
    if (!comm) {
        return MBE_STATUS_INVALID_ARGUMENT;
    }
    DevBytes sums;
    H_TRY(hipMalloc(&sums.p, 4 * sizeof(uint32_t)));
    {
        const int prc = poison_words(static_cast<uint32_t*>(sums.p));
        if (prc < 0) {
            return prc;
        }
    }
    uint32_t out[2] = {0u, 0u};
    const int rc = agree(R, static_cast<ncclComm_t>(comm), value, false, static_cast<uint32_t*>(sums.p), out, static_cast<hipStream_t>(stream));
    if (min_max) {
        min_max[0] = out[0];
        min_max[1] = out[1];
    }
    return rc == MBX_EBADTABLE ? cfail(MBX_EBADTABLE, "mbx_comm_agree: the ranks passed different values") : rc;
}

}  // extern "C"
