/* include/pirip_hip_rccl.h -- the ONE exchange of the multi-GPU receive path, as a C-ABI helper
 * (libpirip_hip_rccl.so; kept out of libpirip_hip.so so that single-GPU users do not pull in RCCL).
 *
 * pirip's IQ channel streams are independent, so they shard one block per GPU with no data-path collective; what a
 * multi-channel receiver exchanges is the decoded bits: each rank's packed bits (pirip_hip_set_bit_packing: 7 bytes per
 * 1200-sample frame at config 1, 0.3 % of the input volume) and its frame counts go to rank 0 in one gather over RCCL
 * (xGMI is point-to-point: a direct send per peer uses each peer's own link; there is no ring to tune).
 * The reference has no counterpart (pipes between processes); this is what BASELINE.json's north_star
 * asks for ("C++ host ... single RCCL gather of decoded bits over xGMI").
 *
 * `nccl_comm` is an ncclComm_t created by the caller (ncclCommInitRank); d_* are device pointers; the call enqueues on
 * `hip_stream` and does not synchronise. */
#ifndef PIRIP_HIP_RCCL_H
#define PIRIP_HIP_RCCL_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* Every rank sends `bytes` bytes from d_send; rank `root` receives world*bytes into d_recv (slot r = rank r's message;
 * its own slot is a device copy). d_recv may be NULL on the other ranks. Returns 0 or a negative PIRIP_ERR_*. */
int pirip_hip_gather_bits(void *nccl_comm, int rank, int world, int root, const void *d_send, size_t bytes, void *d_recv,
                          void *hip_stream);
/* Layout of one rank's gather message, the one the demodulator writes in place (pirip_hip_set_bit_packing(h, 1): d_bits =
 * message, d_nframes = message + *counts_offset): `streams * max_frames * frame_bytes` bytes of packed bits, padding to a
 * 4-byte boundary, then `streams` int32 frame counts. Identical to pirip_amd/shard.py:_payload_layout (tested). */
int pirip_hip_gather_layout(int streams, int64_t max_frames, int frame_bytes, size_t *counts_offset, size_t *total_bytes);
/* rendezvous helper for one-process-per-GPU launches without MPI: rank 0 removes whatever a previous run left at `id_file`,
 * creates the RCCL unique id and publishes it there together with a session tag (written to a temporary name and renamed);
 * the other ranks wait -- up to 60 s ($PIRIP_RCCL_TIMEOUT_S), then PIRIP_ERR_BAD_ARG and a message -- for a file that carries THEIR session tag, so a
 * stale file from a crashed run is never taken for this run's. The tag is $PIRIP_RCCL_SESSION, which the launcher exports
 * fresh for every run and which is REQUIRED when world > 1 (PIRIP_ERR_BAD_ARG without it: nothing the ranks could derive themselves tells
 * two runs from the same shell apart). Then ncclCommInitRank.
 * Returns the communicator through *nccl_comm_out. */
int pirip_hip_rccl_init(const char *id_file, int rank, int world, void **nccl_comm_out);
int pirip_hip_rccl_finalize(void *nccl_comm);
#ifdef __cplusplus
}
#endif
#endif
