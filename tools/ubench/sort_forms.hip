// builder micro-benchmark: how fast does hipCUB group the deletion-variant join's index entries?
//   (a) SortPairs(u32 key, 28 bits; u32 value)   - what the join does
//   (b) SortKeys(u64 = key << 32 | value, bits 32..60)
//   (c) SortPairs with 24 / 20 key bits (what one pass less would buy)
//   (d) (b) through the DoubleBuffer interface
// hipcc --offload-arch=gfx950 -O3 -o sort_forms sort_forms.hip && ./sort_forms [entries]
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(uint32_t* k, uint32_t* v, unsigned long long* c, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)i * 2654435761u; x ^= x >> 15; x *= 0x85EBCA6Bu; x ^= x >> 13;
    k[i] = x & 0x0FFFFFFFu; v[i] = (uint32_t)i; c[i] = (unsigned long long)(x & 0x0FFFFFFFu) << 32 | (uint32_t)i;
}
int main(int argc, char** argv)
{
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 35300000;
    uint32_t *k, *k2, *v, *v2; unsigned long long *c, *c2;
    CK(hipMalloc(&k, n * 4)); CK(hipMalloc(&k2, n * 4)); CK(hipMalloc(&v, n * 4)); CK(hipMalloc(&v2, n * 4)); CK(hipMalloc(&c, n * 8)); CK(hipMalloc(&c2, n * 8));
    size_t t1 = 0, t2 = 0;
    CK(hipcub::DeviceRadixSort::SortPairs(nullptr, t1, k, k2, v, v2, (long long)n, 0, 28));
    CK(hipcub::DeviceRadixSort::SortKeys(nullptr, t2, c, c2, (long long)n, 32, 60));
    void* tmp; CK(hipMalloc(&tmp, t1 > t2 ? t1 : t2));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int form = 0; form < 5; ++form) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, k, v, c, n);
            CK(hipEventRecord(a));
            size_t t = t1 > t2 ? t1 : t2;
            if (form == 0) CK(hipcub::DeviceRadixSort::SortPairs(tmp, t, k, k2, v, v2, (long long)n, 0, 28));
            else if (form == 1) CK(hipcub::DeviceRadixSort::SortKeys(tmp, t, c, c2, (long long)n, 32, 60));
            else if (form == 2) CK(hipcub::DeviceRadixSort::SortPairs(tmp, t, k, k2, v, v2, (long long)n, 4, 28));
            else if (form == 3) CK(hipcub::DeviceRadixSort::SortPairs(tmp, t, k, k2, v, v2, (long long)n, 8, 28));
            else { hipcub::DoubleBuffer<unsigned long long> db(c, c2); CK(hipcub::DeviceRadixSort::SortKeys(tmp, t, db, (long long)n, 32, 60)); }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        const char* names[5] = { "SortPairs u32/u32, 28 bits", "SortKeys u64, bits 32..60", "SortPairs, 24 bits", "SortPairs, 20 bits", "SortKeys u64, DoubleBuffer" };
        printf("%zu entries  %-28s %.3f ms  (%.1f G entries/s)\n", n, names[form], best, n / best / 1e6);
    }
    return 0;
}
