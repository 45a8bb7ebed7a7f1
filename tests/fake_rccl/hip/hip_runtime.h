/* tests/fake_rccl/hip/hip_runtime.h -- TEST-ONLY stand-in for the HIP runtime, seen ONLY by tests/test_gather_fake_nccl.py when
 * it compiles pirip_amd/csrc/rccl_gather.hip as plain C++ on a box without a GPU: "device" pointers are host pointers and the one
 * runtime call the gather makes (the root's own-slot copy) is a memcpy. Never on the product's include path. */
#pragma once
#include <cstddef>
#include <cstring>
typedef void *hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyDeviceToDevice = 3 };
static inline hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t) { memcpy(dst, src, n); return hipSuccess; }
