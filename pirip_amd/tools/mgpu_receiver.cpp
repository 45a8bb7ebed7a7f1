// mgpu_receiver -- C++ host of the multi-GPU receive path (BASELINE config 5): one process per GPU, each
// demodulates its own block of independent IQ channel streams through the C-ABI (no data-path collective), then ONE RCCL
// gather of the packed decoded bits + frame counts to rank 0 (include/pirip_hip_rccl.h). No Python, no torch.
//   RANK / WORLD_SIZE / LOCAL_RANK from the environment (tools/launch_mgpu.sh sets them), rendezvous through a file.
//   mgpu_receiver [--streams B per GPU] [--samples S per stream] [--steps K] [--warmup W] [--id-file PATH]
// Synthetic u8 IQ is generated on each GPU (pirip_hip_synth_cu8: fsk_get_test_bits | fsk_mod -c | u8 quantiser, noise-free),
// resident in HBM before the timed region. Rank 0 prints one JSON line: aggregate IQ Msamples/s over all ranks (barrier by
// the gather itself: rank 0's clock stops when every rank's bits have arrived) and a check that the gathered bits are the
// transmitted test frames.
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/pirip_hip.h"
#include "../../include/pirip_hip_rccl.h"
#include "fsk_plan.hpp"

#define CK(x) do { if ((x) != 0) { fprintf(stderr, "mgpu_receiver[%d]: %s failed at line %d\n", rank, #x, __LINE__); return 2; } } while (0)

int main(int argc, char **argv)
{
    {   // a binary compiled against another header generation must not run against this library (stats rows, stream state sizes)
        const int abi_ok = pirip_hip_abi_check(PIRIP_HIP_ABI_VERSION, PIRIP_STATS_PER_FRAME, sizeof(pirip_stream_state));
        if (!abi_ok) { fprintf(stderr, "%s: built against a different pirip_hip.h than %s\n", argv[0], pirip_hip_version()); return 2; }
    }
    const int rank = getenv("RANK") ? atoi(getenv("RANK")) : 0, world = getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1;
    const int local = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : rank;
    int B = 6144, steps = 5, warmup = 2; long nsamp = 1200000;
    std::string id_file = "/tmp/pirip_rccl_id";
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--streams")) B = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--samples")) nsamp = atol(argv[i + 1]);
        else if (!strcmp(argv[i], "--steps")) steps = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--warmup")) warmup = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--id-file")) id_file = argv[i + 1];
    }
    CK(hipSetDevice(local));
    void *comm = nullptr;
    CK(pirip_hip_rccl_init(id_file.c_str(), rank, world, &comm));
    const int Fs = 240000, Rs = 10000, M = 2, P = 24, Ts = 24, Nsym = 50;
    pirip_fsk_params prm{Fs, Rs, M, P, Nsym, 500, 25000, 0, 100, PIRIP_IN_CU8_FSKDEMOD};
    pirip_hip_demod *h = nullptr;
    CK(pirip_hip_create(&prm, B, local, &h));
    CK(pirip_hip_set_bit_packing(h, 1));
    // test bits (the 100-bit frame repeated), one tone plan per stream, timing offsets 0..23
    const long nsym = nsamp / Ts + 2 * Nsym;
    std::vector<uint8_t> frame(100), txbits((size_t)nsym);
    pirip::test_frame_bits(frame.data(), 100);
    for (long i = 0; i < nsym; i++) txbits[(size_t)i] = frame[(size_t)(i % 100)];
    std::vector<int32_t> f1((size_t)B), skip((size_t)B);
    for (int s = 0; s < B; s++) { const long g = (long)rank * B + s; f1[(size_t)s] = 10000 + (int)((g % 5) - 2) * 937; skip[(size_t)s] = (int)((g / 5) % Ts); }
    uint8_t *d_tx = nullptr, *d_iq = nullptr, *d_msg[2] = {nullptr, nullptr}, *d_all = nullptr;
    const long maxf = nsamp / (Ts * Nsym - Ts / 4) + 2;
    const size_t frame_bytes = 7;
    size_t cnt_off = 0, msg = 0;                     // packed bits | pad to 4 bytes | int32 frame counts (pirip_hip_gather_layout)
    CK(pirip_hip_gather_layout(B, maxf, (int)frame_bytes, &cnt_off, &msg));
    CK(hipMalloc((void **)&d_tx, (size_t)nsym));
    CK(hipMalloc((void **)&d_iq, (size_t)B * nsamp * 2));
    // two messages alternate: the gather of step i runs on its own stream under the demodulator launch of step i + 1
    CK(hipMalloc((void **)&d_msg[0], msg));
    CK(hipMalloc((void **)&d_msg[1], msg));
    if (rank == 0) CK(hipMalloc((void **)&d_all, msg * world));
    CK(hipMemcpy(d_tx, txbits.data(), (size_t)nsym, hipMemcpyHostToDevice));
    CK(pirip_hip_synth_cu8(Fs, Rs, M, B, f1.data(), 10000, skip.data(), d_tx, 0, nsym, d_iq, (size_t)nsamp * 2, nsamp, 32.0f, 0.0f, 1, nullptr));
    int64_t *d_cons = nullptr;
    CK(hipMalloc((void **)&d_cons, sizeof(int64_t) * (size_t)B));
    hipStream_t s_dem = nullptr, s_com = nullptr;
    hipEvent_t demod_done[2], gather_done[2];
    CK(hipStreamCreateWithFlags(&s_dem, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_com, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) { CK(hipEventCreateWithFlags(&demod_done[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&gather_done[i], hipEventDisableTiming)); }
    long nstep = 0;
    auto step = [&]() -> int {
        const int k = (int)(nstep & 1);
        uint8_t *m = d_msg[k];
        if (nstep >= 2 && hipStreamWaitEvent(s_dem, gather_done[k], 0) != hipSuccess) return 1;     // this message's last gather has been sent
        if (pirip_hip_demod_batch(h, d_iq, (size_t)nsamp * 2, nsamp, m, (size_t)maxf * frame_bytes, nullptr, 0, nullptr, 0,
                                  (int32_t *)(m + cnt_off), d_cons, maxf, s_dem)) return 1;
        if (hipEventRecord(demod_done[k], s_dem) != hipSuccess || hipStreamWaitEvent(s_com, demod_done[k], 0) != hipSuccess) return 1;
        if (pirip_hip_gather_bits(comm, rank, world, 0, m, msg, d_all, s_com)) return 1;
        nstep++;
        return hipEventRecord(gather_done[k], s_com) != hipSuccess;
    };
    for (int i = 0; i < warmup; i++) CK(step());
    CK(hipDeviceSynchronize());
    // rendezvous before the clock starts: a zero-payload round of the same gather
    CK(pirip_hip_gather_bits(comm, rank, world, 0, d_msg[0], 8, d_all, s_com));
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; i++) CK(step());
    CK(hipDeviceSynchronize());
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rank == 0) {
        // every rank's slot: frame counts and the first stream's bits must be the test frames (after the estimators settle)
        std::vector<uint8_t> all(msg * (size_t)world);
        CK(hipMemcpy(all.data(), d_all, all.size(), hipMemcpyDeviceToHost));
        long frames = 0, bad = 0, checked = 0;
        for (int r = 0; r < world; r++) {
            const uint8_t *m = all.data() + (size_t)r * msg;
            const int32_t *nf = (const int32_t *)(m + cnt_off);
            for (int s = 0; s < B; s++) frames += nf[s];
            pirip::PutBits pb; pb.init(100, 0.1f);
            for (long f = 20; f < nf[0]; f++)
                for (int b = 0; b < Nsym; b++) { int e; pb.push((m[(size_t)f * frame_bytes + (size_t)(b >> 3)] >> (7 - (b & 7))) & 1, &e); }
            bad += pb.biterr; checked += pb.bitcnt;
        }
        const double samples = (double)frames * Ts * Nsym;
        printf("{\"metric\": \"IQ Msamples/s demodulated (2-FSK Fs=240k Rs=10k), C++ host + RCCL gather\", \"value\": %.1f, \"n_gpus\": %d, "
               "\"steps\": %d, \"ms_per_step\": %.3f, \"streams_per_gpu\": %d, \"frames_gathered_per_step\": %ld, \"gather_bytes_per_rank\": %zu, "
               "\"test_bits_checked\": %ld, \"bit_errors_vs_tx\": %ld}\n",
               samples * steps / dt / 1e6, world, steps, dt / steps * 1e3, B, frames, msg, checked, bad);
    }
    pirip_hip_destroy(h);
    pirip_hip_rccl_finalize(comm);
    if (rank == 0) unlink(id_file.c_str());
    return 0;
}
