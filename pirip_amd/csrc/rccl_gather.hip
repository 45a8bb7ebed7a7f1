// pirip_amd/csrc/rccl_gather.hip -- include/pirip_hip_rccl.h: the single packed-bits gather of the multi-GPU path.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/pirip_hip.h"
#include "../../include/pirip_hip_rccl.h"

extern "C" {

int pirip_hip_gather_bits(void *nccl_comm, int rank, int world, int root, const void *d_send, size_t bytes, void *d_recv, void *hip_stream)
{
    if (!nccl_comm || !d_send || world <= 0 || rank < 0 || rank >= world || root < 0 || root >= world) return PIRIP_ERR_BAD_ARG;
    if (rank == root && !d_recv) return PIRIP_ERR_BAD_ARG;
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    hipStream_t st = (hipStream_t)hip_stream;
    if (rank == root) {
        // own slot: a device copy; every other slot: one receive straight from that peer (its own xGMI link)
        if (hipMemcpyAsync((char *)d_recv + (size_t)root * bytes, d_send, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return PIRIP_ERR_HIP;
        if (world == 1) return PIRIP_OK;
        if (ncclGroupStart() != ncclSuccess) return PIRIP_ERR_HIP;
        for (int r = 0; r < world; r++)
            if (r != root && ncclRecv((char *)d_recv + (size_t)r * bytes, bytes, ncclUint8, r, comm, st) != ncclSuccess) { ncclGroupEnd(); return PIRIP_ERR_HIP; }
        return ncclGroupEnd() == ncclSuccess ? PIRIP_OK : PIRIP_ERR_HIP;
    }
    return ncclSend(d_send, bytes, ncclUint8, root, comm, st) == ncclSuccess ? PIRIP_OK : PIRIP_ERR_HIP;
}

int pirip_hip_rccl_init(const char *id_file, int rank, int world, void **out)
{
    if (!id_file || !out || world <= 0 || rank < 0 || rank >= world) return PIRIP_ERR_BAD_ARG;
    ncclUniqueId id;
    if (rank == 0) {
        if (ncclGetUniqueId(&id) != ncclSuccess) return PIRIP_ERR_HIP;
        const std::string tmp = std::string(id_file) + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(&id, sizeof(id), 1, f) != 1) { if (f) fclose(f); return PIRIP_ERR_BAD_ARG; }
        fclose(f);
        if (rename(tmp.c_str(), id_file) != 0) return PIRIP_ERR_BAD_ARG;
    } else {
        FILE *f = nullptr;
        for (int tries = 0; tries < 6000 && !(f = fopen(id_file, "rb")); tries++) usleep(10000);
        if (!f || fread(&id, sizeof(id), 1, f) != 1) { if (f) fclose(f); return PIRIP_ERR_BAD_ARG; }
        fclose(f);
    }
    ncclComm_t comm;
    if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess) return PIRIP_ERR_HIP;
    *out = (void *)comm;
    return PIRIP_OK;
}

int pirip_hip_rccl_finalize(void *nccl_comm)
{
    if (!nccl_comm) return PIRIP_ERR_BAD_ARG;
    return ncclCommDestroy((ncclComm_t)nccl_comm) == ncclSuccess ? PIRIP_OK : PIRIP_ERR_HIP;
}

}  // extern "C"
