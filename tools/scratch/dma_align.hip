// scratch: does buffer_load_dwordx4 ... lds accept global byte offsets that are only 4- or 8-byte aligned? what do out-of-range lanes write?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned char* in, unsigned* out, int nbytes, int off)
{
    __shared__ __attribute__((aligned(16))) unsigned raw[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) raw[i] = 0xdeadbeefu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)raw, 16, threadIdx.x*16, off, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(raw+256), 4, threadIdx.x*4, off, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = raw[i];
}
int main()
{
    const int nb = 1000;   // bytes valid
    std::vector<unsigned char> h(4096);
    for (int i = 0; i < 4096; i++) h[i] = (unsigned char)(i * 7 + 3);
    unsigned char* d; unsigned* o;
    hipMalloc(&d, 4096); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    for (int off : {0, 4, 8, 12, 2}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, nb, off);
        std::vector<unsigned> r(1024);
        if (hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost) != hipSuccess) { printf("off %d: error %s\n", off, hipGetErrorString(hipGetLastError())); continue; }
        int bad16 = 0, bad4 = 0, oob16 = -1;
        for (int i = 0; i < 256; i++) {
            unsigned exp = 0; bool in = (off + 4 * i + 4 <= nb);
            for (int b = 0; b < 4; b++) exp |= (unsigned)h[off + 4 * i + b] << (8 * b);
            if (in && r[i] != exp) bad16++;
            if (!in && oob16 < 0) oob16 = i;
        }
        for (int i = 0; i < 64; i++) {
            unsigned exp = 0; for (int b = 0; b < 4; b++) exp |= (unsigned)h[off + 4 * i + b] << (8 * b);
            if (r[256 + i] != exp) bad4++;
        }
        printf("offset %2d: x4 mismatches %d (first out-of-range dword %d holds %08x, last dword %08x), dword mismatches %d\n", off, bad16, oob16, oob16 >= 0 ? r[oob16] : 0, r[255], bad4);
    }
    return 0;
}
