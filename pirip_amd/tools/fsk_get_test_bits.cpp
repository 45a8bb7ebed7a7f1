// fsk_get_test_bits -- emit the repeating 100-bit test frame, one bit per byte.
// CLI of codec2's tool of the same name [UPSTREAM-RECALLED codec2 src/fsk_get_test_bits.c]:
//   fsk_get_test_bits OutputBitsOnePerByte numBits [frameLengthBits]
// argv pinned by /root/reference/README.md:101,179 and test/loopback_rtl_fsk.sh:17. CPU tool
// (Tx side / measurement instrument), not on the GPU path.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fsk_plan.hpp"

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s OutputBitsOnePerByte numBits [frameLengthBits]\n", argv[0]);
        return 1;
    }
    FILE *fout = strcmp(argv[1], "-") ? fopen(argv[1], "wb") : stdout;
    if (!fout) { fprintf(stderr, "Couldn't open output file: %s\n", argv[1]); return 1; }
    long nbits = atol(argv[2]);
    int framesize = argc > 3 ? atoi(argv[3]) : 100;
    if (framesize <= 0) return 1;
    std::vector<uint8_t> frame(framesize);
    pirip::test_frame_bits(frame.data(), framesize);
    // whole frames only, as upstream does: ceil(numBits/framesize) frames
    for (long sent = 0; sent < nbits; sent += framesize) {
        if (fwrite(frame.data(), 1, framesize, fout) != (size_t)framesize) return 1;
        if (fout == stdout) fflush(fout);
    }
    if (fout != stdout) fclose(fout);
    return 0;
}
