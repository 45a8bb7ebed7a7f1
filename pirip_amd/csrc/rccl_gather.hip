// pirip_amd/csrc/rccl_gather.hip -- include/pirip_hip_rccl.h: the single packed-bits gather of the multi-GPU path.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
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

int pirip_hip_gather_layout(int streams, int64_t max_frames, int frame_bytes, size_t *counts_offset, size_t *total_bytes)
{
    if (streams <= 0 || max_frames < 0 || frame_bytes <= 0 || !counts_offset || !total_bytes) return PIRIP_ERR_BAD_ARG;
    const size_t nb = (size_t)streams * (size_t)max_frames * (size_t)frame_bytes;
    *counts_offset = (nb + 3) / 4 * 4;                       // int32 stores must be aligned whatever the stream count
    *total_bytes = *counts_offset + sizeof(int32_t) * (size_t)streams;
    return PIRIP_OK;
}

namespace {
struct IdFile { char magic[8]; char session[56]; ncclUniqueId id; };
// The tag that tells this run's unique-id file from one a crashed run left behind. It has to come from the launcher
// ($PIRIP_RCCL_SESSION: tools/launch_mgpu.sh exports a fresh one per run): anything the ranks could derive by themselves -- the
// parent's pid was the round-3 fallback -- is the same for two runs started by hand from one shell, and a rank > 0 would then
// take the stale file of the earlier run and hang in ncclCommInitRank with the wrong id.
bool session_tag(char out[56])
{
    const char *e = getenv("PIRIP_RCCL_SESSION");
    if (!e || !*e) return false;
    snprintf(out, 56, "%s", e);
    return true;
}
}  // namespace

int pirip_hip_rccl_init(const char *id_file, int rank, int world, void **out)
{
    if (!id_file || !out || world <= 0 || rank < 0 || rank >= world) return PIRIP_ERR_BAD_ARG;
    IdFile rec;
    memset(&rec, 0, sizeof(rec));
    char want[56] = "single";
    if (!session_tag(want) && world > 1) {
        fprintf(stderr, "pirip_hip_rccl_init: world size %d needs $PIRIP_RCCL_SESSION (one fresh value per run, the same on every rank; "
                        "tools/launch_mgpu.sh sets it)\n", world);
        return PIRIP_ERR_BAD_ARG;
    }
    if (rank == 0) {
        (void)unlink(id_file);                                // whatever a crashed run left behind is not ours
        memcpy(rec.magic, "PIRIPID1", 8);
        memcpy(rec.session, want, sizeof(want));
        if (ncclGetUniqueId(&rec.id) != ncclSuccess) return PIRIP_ERR_HIP;
        const std::string tmp = std::string(id_file) + ".tmp." + std::to_string((long)getpid());
        FILE *f = fopen(tmp.c_str(), "wbx");                  // exclusive create: never follows a planted link
        if (!f || fwrite(&rec, sizeof(rec), 1, f) != 1) { if (f) fclose(f); (void)unlink(tmp.c_str()); return PIRIP_ERR_BAD_ARG; }
        fclose(f);
        if (rename(tmp.c_str(), id_file) != 0) { (void)unlink(tmp.c_str()); return PIRIP_ERR_BAD_ARG; }
    } else {
        bool ok = false;
        const char *te = getenv("PIRIP_RCCL_TIMEOUT_S");
        const int max_tries = 100 * (te && atoi(te) > 0 ? atoi(te) : 60);
        for (int tries = 0; tries < max_tries && !ok; tries++) {   // 60 s unless $PIRIP_RCCL_TIMEOUT_S says otherwise
            FILE *f = fopen(id_file, "rb");
            if (f) {
                ok = fread(&rec, sizeof(rec), 1, f) == 1 && !memcmp(rec.magic, "PIRIPID1", 8) && !strncmp(rec.session, want, sizeof(want));
                fclose(f);
            }
            if (!ok) usleep(10000);
        }
        if (!ok) {
            fprintf(stderr, "pirip_hip_rccl_init: rank %d: no unique-id file for session '%s' at %s in time (is rank 0 running? a file "
                            "from another run is ignored)\n", rank, want, id_file);
            return PIRIP_ERR_BAD_ARG;
        }
    }
    const ncclUniqueId id = rec.id;
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
