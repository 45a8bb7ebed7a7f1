// tools/hbm_read_ceiling.hip -- measurement aid (not part of the product): the read-only HBM ceiling of one
// MI355X for fully coalesced 16-byte loads, to put the decimator's and the demodulator's input rates in context.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_read tools/hbm_read_ceiling.hip && /tmp/hbm_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const uint4 *in, size_t n16, uint32_t *out)
{
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n16; i += stride) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = (i + (size_t)u * 256 < n16) ? in[i + (size_t)u * 256] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;   // practically never: keeps the loads alive without a write stream
}

int main()
{
    const size_t bytes = (size_t)6 << 30;   // 6 GiB: far larger than the 256 MiB Infinity Cache
    uint4 *d; uint32_t *o;
    if (hipMalloc((void **)&d, bytes) != hipSuccess || hipMalloc((void **)&o, 1 << 22) != hipSuccess) return 1;
    hipMemset(d, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 128};
    for (int gi = 0; gi < 5; gi++) {
        for (int un = 0; un < 3; un++) {
            const int grid = grids[gi];
            auto launch = [&]() {
                if (un == 0) hipLaunchKernelGGL(read_kernel<2>, dim3(grid), dim3(256), 0, 0, d, bytes / 16, o);
                else if (un == 1) hipLaunchKernelGGL(read_kernel<4>, dim3(grid), dim3(256), 0, 0, d, bytes / 16, o);
                else hipLaunchKernelGGL(read_kernel<8>, dim3(grid), dim3(256), 0, 0, d, bytes / 16, o);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            for (int it = 0; it < 5; it++) launch();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            printf("grid %6d x256 unroll %d: %.3f ms  %.0f GB/s\n", grid, un == 0 ? 2 : un == 1 ? 4 : 8, ms, bytes / ms / 1e6);
        }
    }
    return 0;
}
