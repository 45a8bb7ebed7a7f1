#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k(float* out)
{
    const int lane = threadIdx.x;
    float mag[16];
    for (int u = 0; u < 16; u++) { v2f t{(float)(lane * 100 + u), 1.0f}; t = t * v2f{1.0f, 2.0f}; mag[u] = t.x + 0.0f * t.y; }
    float A[8], B[8];
    for (int u = 0; u < 8; u++) {
        float xa = mag[u], xb = mag[u + 8]; asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(xa), "+v"(xb)); const float r[2] = {xa, xb};
        A[u] = r[0]; B[u] = r[1];
    }
    for (int u = 0; u < 8; u++) { out[(lane * 8 + u) * 2] = A[u]; out[(lane * 8 + u) * 2 + 1] = B[u]; }
}
int main()
{
    float* d; hipMalloc(&d, 64 * 16 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[1024]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int lane : {0, 5, 32, 37}) { printf("lane %2d:", lane); for (int u = 0; u < 3; u++) printf("  A[%d]=%6.0f B[%d]=%6.0f", u, h[(lane * 8 + u) * 2], u, h[(lane * 8 + u) * 2 + 1]); printf("\n"); }
    printf("expect lane 5: A[u]=500+u (own, u) B[u]=3700+u (lane 37's u);  lane 37: A[u]=500+u+8 (lane 5's u+8) B[u]=3700+u+8 (own)\n");
    return 0;
}
