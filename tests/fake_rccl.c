/* fake_rccl.c -- TEST INFRASTRUCTURE, not product code and not RCCL: the eight entry points of the RCCL API that
 * mbelib-neo_amd/csrc/mbx_collective.hip binds (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclCommCount,
 * ncclCommUserRank, ncclBroadcast, ncclAllReduce, ncclGetErrorString), implemented for RANKS THAT ARE HOST THREADS OF ONE
 * PROCESS SHARING ONE DEVICE: a collective is a rendezvous of the threads (mutex + condition variable) around plain
 * device-to-device / device-to-host copies.  Real RCCL refuses two ranks on one device, and no box this work has ever run on
 * had more than one GPU, so the N > 1 control flow of mbx_init_broadcast / mbx_comm_agree -- non-root ranks receiving the
 * blob, a rank whose tables differ, a rank that fails on the way -- had never executed.  With MBX_RCCL_LIBRARY pointing at
 * this library it runs with 2, 4 and 8 thread-ranks on the one GPU of a test box (tests/test_gpu_parity.py).
 *
 * Semantics kept from RCCL: calls are made by every rank in the same order; data movement is ordered after the work already
 * queued on the caller's stream (the stream is synchronised first) and complete when the call returns (stricter than RCCL,
 * which is asynchronous; the caller synchronises anyway).  Fault injection (environment, read at communicator creation):
 *   FAKE_RCCL_CORRUPT_RANK=k     rank k receives the broadcast with one byte flipped (a transport error mbx_init must catch)
 *   FAKE_RCCL_FAIL_BCAST_RANK=k  ncclBroadcast returns ncclSystemError on rank k (which still meets the others: the fake models
 *                                a LOCAL failure, not a dead rank)
 * Build: gcc -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/fake_rccl.c -L/opt/rocm/lib -lamdhip64 -lpthread
 */
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;

#define MAX_RANKS 16
typedef struct Group {
    unsigned long long id;
    int nranks, joined, refs;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    /* one collective at a time: ranks deposit, the last one in combines, everybody leaves with the result */
    int arrived, generation;
    const void* ptr[MAX_RANKS];
    uint32_t word[MAX_RANKS], result;
    int corrupt_rank, fail_bcast_rank;
    struct Group* next;
} Group;
typedef struct Comm {
    Group* g;
    int rank;
}* ncclComm_t;

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static Group* g_groups = NULL;
static unsigned long long g_next_id = 1;

static int env_int(const char* name) {
    const char* e = getenv(name);
    return (e && e[0]) ? atoi(e) : -1;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) {
        return ncclInvalidArgument;
    }
    memset(id, 0, sizeof(*id));
    pthread_mutex_lock(&g_mu);
    const unsigned long long v = g_next_id++;
    pthread_mutex_unlock(&g_mu);
    memcpy(id->internal, "FAKERCCL", 8);
    memcpy(id->internal + 8, &v, sizeof(v));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks || memcmp(id.internal, "FAKERCCL", 8) != 0) {
        return ncclInvalidArgument;
    }
    unsigned long long v;
    memcpy(&v, id.internal + 8, sizeof(v));
    pthread_mutex_lock(&g_mu);
    Group* g = g_groups;
    while (g && g->id != v) {
        g = g->next;
    }
    if (!g) {
        g = (Group*)calloc(1, sizeof(Group));
        g->id = v;
        g->nranks = nranks;
        pthread_mutex_init(&g->mu, NULL);
        pthread_cond_init(&g->cv, NULL);
        g->corrupt_rank = env_int("FAKE_RCCL_CORRUPT_RANK");
        g->fail_bcast_rank = env_int("FAKE_RCCL_FAIL_BCAST_RANK");
        g->next = g_groups;
        g_groups = g;
    }
    g->refs++;
    pthread_mutex_unlock(&g_mu);
    if (g->nranks != nranks) {
        return ncclInvalidArgument;
    }
    /* like the real thing, the call returns when every rank has joined */
    pthread_mutex_lock(&g->mu);
    g->joined++;
    pthread_cond_broadcast(&g->cv);
    while (g->joined < g->nranks) {
        pthread_cond_wait(&g->cv, &g->mu);
    }
    pthread_mutex_unlock(&g->mu);
    struct Comm* c = (struct Comm*)calloc(1, sizeof(struct Comm));
    c->g = g;
    c->rank = rank;
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (comm) {
        pthread_mutex_lock(&g_mu);
        comm->g->refs--;   /* groups are small and a test process is short-lived: they are not unlinked */
        pthread_mutex_unlock(&g_mu);
        free(comm);
    }
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    if (!comm || !count) {
        return ncclInvalidArgument;
    }
    *count = comm->g->nranks;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
    if (!comm || !rank) {
        return ncclInvalidArgument;
    }
    *rank = comm->rank;
    return ncclSuccess;
}

/* every rank deposits (pointer, word); when the last one is in, `combine` (if any) runs once; returns with all deposits visible.
 * leave = 0: the rendezvous BEFORE the data movement; call again with leave = 1 after it so that nobody's buffer goes away early */
static void meet(Group* g, int rank, const void* p, uint32_t w, ncclRedOp_t op, int reduce, int deposit) {
    pthread_mutex_lock(&g->mu);
    const int gen = g->generation;
    if (deposit) {   /* (the rendezvous AFTER a data movement deposits nothing: a fast rank must not wipe what a slow one still reads) */
        g->ptr[rank] = p;
        g->word[rank] = w;
    }
    if (++g->arrived == g->nranks) {
        if (reduce) {
            uint32_t r = g->word[0];
            for (int i = 1; i < g->nranks; ++i) {
                r = (op == ncclMin) ? (g->word[i] < r ? g->word[i] : r) : (g->word[i] > r ? g->word[i] : r);
            }
            g->result = r;
        }
        g->arrived = 0;
        g->generation++;
        pthread_cond_broadcast(&g->cv);
    } else {
        while (g->generation == gen) {
            pthread_cond_wait(&g->cv, &g->mu);
        }
    }
    pthread_mutex_unlock(&g->mu);
}

ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, int root, ncclComm_t comm, hipStream_t stream) {
    if (!comm || !recvbuff || root < 0 || root >= comm->g->nranks || (type != ncclUint8 && type != ncclInt8)) {
        return ncclInvalidArgument;
    }
    Group* g = comm->g;
    ncclResult_t rc = ncclSuccess;
    if (hipStreamSynchronize(stream) != hipSuccess) {   /* the root's upload is queued on its stream */
        rc = ncclUnhandledCudaError;
    }
    meet(g, comm->rank, comm->rank == root ? sendbuff : recvbuff, 0u, ncclMin, 0, 1);
    if (comm->rank != root && rc == ncclSuccess) {
        if (hipMemcpy(recvbuff, g->ptr[root], count, hipMemcpyDeviceToDevice) != hipSuccess) {
            rc = ncclUnhandledCudaError;
        }
        if (comm->rank == g->corrupt_rank && count > 100) {   /* a flipped byte somewhere in the tables */
            unsigned char b;
            if (hipMemcpy(&b, (const char*)recvbuff + 100, 1, hipMemcpyDeviceToHost) == hipSuccess) {
                b ^= 0x40;
                (void)hipMemcpy((char*)recvbuff + 100, &b, 1, hipMemcpyHostToDevice);
            }
        }
    }
    meet(g, comm->rank, NULL, 0u, ncclMin, 0, 0);   /* the root's buffer stays until everybody has copied */
    if (comm->rank == g->fail_bcast_rank) {
        rc = ncclSystemError;
    }
    return rc;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
    if (!comm || !sendbuff || !recvbuff || count != 1 || type != ncclUint32 || (op != ncclMin && op != ncclMax)) {
        return ncclInvalidArgument;   /* all mbx_collective.hip ever asks for */
    }
    Group* g = comm->g;
    ncclResult_t rc = ncclSuccess;
    uint32_t w = 0u;
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(&w, sendbuff, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) {
        rc = ncclUnhandledCudaError;
        w = (op == ncclMin) ? 0u : 0xffffffffu;   /* a rank that cannot read its value cannot agree */
    }
    meet(g, comm->rank, NULL, w, op, 1, 1);
    const uint32_t r = g->result;
    meet(g, comm->rank, NULL, 0u, op, 0, 0);     /* everybody has read the result before the next collective overwrites it */
    if (hipMemcpy(recvbuff, &r, sizeof(r), hipMemcpyHostToDevice) != hipSuccess) {
        rc = ncclUnhandledCudaError;
    }
    return rc;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake RCCL)";
    case ncclSystemError: return "injected system error (fake RCCL)";
    case ncclInvalidArgument: return "invalid argument (fake RCCL)";
    default: return "internal error (fake RCCL)";
    }
}
