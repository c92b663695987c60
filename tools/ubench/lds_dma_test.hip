// Micro-test of two gfx950 instructions the scan kernel relies on through inline assembly:
//   global_load_lds_dwordx4  (where does lane l's 16 bytes land in LDS?)
//   ds_read_u8_d16 / ds_read_u8_d16_hi  (which register half is written, is the other half preserved?)
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_dma_test tools/ubench/lds_dma_test.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_dma(const uint8_t* __restrict__ src, uint32_t* __restrict__ out, int reverse)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_buf[2][1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) reinterpret_cast<uint32_t*>(s_buf)[i] = 0xDEADBEEFu;
    __syncthreads();
    const uint32_t ldsbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(&s_buf[1][0]);
    const uint8_t* g = src + 16 * (reverse ? 63 - lane : lane);
    const uint32_t sbase = __builtin_amdgcn_readfirstlane(ldsbase);
    asm volatile("s_mov_b32 m0, %0\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "s_waitcnt vmcnt(0)" :: "s"(sbase), "v"(g) : "m0", "memory");
    __syncthreads();
    const uint4 v = *reinterpret_cast<const uint4*>(&s_buf[1][16 * lane]);
    out[4 * lane + 0] = v.x; out[4 * lane + 1] = v.y; out[4 * lane + 2] = v.z; out[4 * lane + 3] = v.w;
    if (lane == 0) out[256] = reinterpret_cast<uint32_t*>(s_buf)[0];      // buffer 0 untouched?
}

__global__ void k_d16(uint32_t* __restrict__ out)
{
    __shared__ uint8_t s_tab[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) s_tab[i] = (uint8_t)(i ^ 0xA5);
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)s_tab;
    uint32_t r = 0x12345678u;
    const uint32_t a0 = base + lane, a1 = base + 128 + lane;
    asm volatile("ds_read_u8_d16 %0, %1\n\t"
                 "ds_read_u8_d16_hi %0, %2\n\t"
                 "s_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(a0), "v"(a1));
    out[lane] = r;
    uint32_t q = 0x12345678u;
    asm volatile("ds_read_u8_d16_hi %0, %1\n\t"
                 "s_waitcnt lgkmcnt(0)" : "+v"(q) : "v"(a0));
    out[64 + lane] = q;
}

int main()
{
    std::vector<uint8_t> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (uint8_t)(i * 7 + (i >> 4));
    uint8_t* d; uint32_t* o;
    hipMalloc(&d, 1024); hipMalloc(&o, 4 * 512);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    std::vector<uint32_t> r(512);
    for (int rev = 0; rev < 2; ++rev) {
        hipLaunchKernelGGL(k_dma, dim3(1), dim3(64), 0, 0, d, o, rev);
        hipMemcpy(r.data(), o, 4 * 512, hipMemcpyDeviceToHost);
        int ok_lane = 1, ok_other = 1;
        for (int l = 0; l < 64; ++l) for (int b = 0; b < 16; ++b) {
            const uint8_t got = reinterpret_cast<uint8_t*>(r.data())[16 * l + b];
            const int srcl = rev ? 63 - l : l;
            if (got != h[16 * srcl + b]) ok_lane = 0;
        }
        if (r[256] != 0xDEADBEEFu) ok_other = 0;
        printf("global_load_lds_dwordx4 reverse=%d: LDS[M0 + 16*lane] = lane's 16 bytes: %s; other buffer untouched: %s\n", rev, ok_lane ? "yes" : "NO", ok_other ? "yes" : "NO");
        if (!ok_lane) { printf("  first words:"); for (int i = 0; i < 8; ++i) printf(" %08x", r[i]); printf("\n"); }
    }
    hipLaunchKernelGGL(k_d16, dim3(1), dim3(64), 0, 0, o);
    hipMemcpy(r.data(), o, 4 * 128, hipMemcpyDeviceToHost);
    int ok = 1, okq = 1;
    for (int l = 0; l < 64; ++l) {
        const uint32_t want = (uint32_t)((l ^ 0xA5) & 0xFF) | ((uint32_t)(((128 + l) ^ 0xA5) & 0xFF) << 16);
        if (r[l] != want) ok = 0;
        const uint32_t wq = 0x5678u | ((uint32_t)((l ^ 0xA5) & 0xFF) << 16);
        if (r[64 + l] != wq) okq = 0;
    }
    printf("ds_read_u8_d16 + _d16_hi into one register = lo | hi << 16 (zero-extended bytes): %s (lane 1: %08x)\n", ok ? "yes" : "NO", r[1]);
    printf("ds_read_u8_d16_hi keeps the low half: %s (lane 1: %08x)\n", okq ? "yes" : "NO", r[65]);
    return 0;
}
