/* tests/fake_rccl/rccl/rccl.h -- TEST-ONLY: the handful of RCCL entry points pirip_amd/csrc/rccl_gather.hip calls, implemented by
 * tests/fake_rccl/fake_nccl.cpp over named pipes between processes (one FIFO per ordered rank pair under $FAKE_NCCL_DIR), so that
 * the C++ gather -- slot layout, group semantics, rendezvous -- runs at world size 2 on a box without GPUs. */
#pragma once
#include <cstddef>
typedef struct fakeComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclSystemError = 2 } ncclResult_t;
typedef enum { ncclUint8 = 1 } ncclDataType_t;
extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id);
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, void *stream);
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, void *stream);
}
