// tests/cprog/gather_world2.cpp -- TEST-ONLY driver of the C++ multi-GPU exchange at world size > 1 without GPUs: built by
// tests/test_gather_fake_nccl.py from pirip_amd/csrc/rccl_gather.hip (the product's source, compiled as plain C++ against
// tests/fake_rccl/) + tests/fake_rccl/fake_nccl.cpp. Each rank fills a gather message laid out by pirip_hip_gather_layout with a
// pattern keyed by (rank, stream, frame, byte), meets the others through pirip_hip_rccl_init's file rendezvous and calls
// pirip_hip_gather_bits twice (two alternating messages, as mgpu_receiver does); rank 0 writes what it gathered to stdout.
//   gather_world2 <rank> <world> <id_file> <streams> <max_frames> <frame_bytes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pirip_hip.h"
#include "pirip_hip_rccl.h"

int main(int argc, char **argv)
{
    if (argc < 7) return 2;
    const int rank = atoi(argv[1]), world = atoi(argv[2]), streams = atoi(argv[4]), maxf = atoi(argv[5]), fb = atoi(argv[6]);
    size_t off = 0, total = 0;
    if (pirip_hip_gather_layout(streams, maxf, fb, &off, &total) != PIRIP_OK) return 3;
    void *comm = nullptr;
    if (pirip_hip_rccl_init(argv[3], rank, world, &comm) != PIRIP_OK) return 4;
    std::vector<unsigned char> all(rank == 0 ? total * (size_t)world : 1);
    for (int round = 0; round < 2; round++) {
        std::vector<unsigned char> msg(total, 0xEE);            // padding bytes keep the fill value: the parser must not read them
        for (int s = 0; s < streams; s++)
            for (int f = 0; f < maxf; f++)
                for (int b = 0; b < fb; b++) msg[((size_t)s * maxf + f) * fb + b] = (unsigned char)(17 * rank + 5 * s + 3 * f + b + 101 * round);
        for (int s = 0; s < streams; s++) { const int32_t n = (rank + 1) * 1000 + s + round; memcpy(&msg[off + 4 * (size_t)s], &n, 4); }
        if (pirip_hip_gather_bits(comm, rank, world, 0, msg.data(), total, rank == 0 ? all.data() : nullptr, nullptr) != PIRIP_OK) return 5;
    }
    if (rank == 0) fwrite(all.data(), 1, all.size(), stdout);
    pirip_hip_rccl_finalize(comm);
    return 0;
}
