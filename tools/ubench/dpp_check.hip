// Checks on the GPU that the gfx9 wave-wide DPP controls used by the kernels behave as documented on gfx950:
//   wave_shl:1 (lane i <- lane i+1, lane 63 <- 0), wave_shr:1 (lane i <- lane i-1, lane 0 <- 0) and the
//   row_shr / row_bcast inclusive scan.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/dpp_check tools/ubench/dpp_check.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint32_t* in, uint32_t* shl, uint32_t* scan, uint32_t* shr, int nact)
{
    const int lane = threadIdx.x;
    uint32_t x = in[lane];
    uint32_t a = 0xDEADu, s = 0, r = 0xDEADu;
    if (lane < nact) {      // partially active wave: inactive source lanes must read as 0 (bound_ctrl)
        a = __builtin_amdgcn_update_dpp(0u, x, 0x130, 0xF, 0xF, true);
        r = __builtin_amdgcn_update_dpp(0u, x, 0x138, 0xF, 0xF, true);
        s = x;
        s += __builtin_amdgcn_update_dpp(0u, s, 0x111, 0xF, 0xF, true);
        s += __builtin_amdgcn_update_dpp(0u, s, 0x112, 0xF, 0xF, true);
        s += __builtin_amdgcn_update_dpp(0u, s, 0x114, 0xF, 0xF, true);
        s += __builtin_amdgcn_update_dpp(0u, s, 0x118, 0xF, 0xF, true);
        s += __builtin_amdgcn_update_dpp(0u, s, 0x142, 0xA, 0xF, true);
        s += __builtin_amdgcn_update_dpp(0u, s, 0x143, 0xC, 0xF, true);
    }
    shl[lane] = a; scan[lane] = s; shr[lane] = r;
}
int main()
{
    uint32_t h[64], a[64], s[64], r[64], *d, *da, *ds, *dr;
    for (int i = 0; i < 64; ++i) h[i] = 1000u + 7u * i * i;
    hipMalloc(&d, 256); hipMalloc(&da, 256); hipMalloc(&ds, 256); hipMalloc(&dr, 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    int bad = 0;
    for (int nact : { 64, 40 }) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, da, ds, dr, nact);
        hipMemcpy(a, da, 256, hipMemcpyDeviceToHost); hipMemcpy(s, ds, 256, hipMemcpyDeviceToHost); hipMemcpy(r, dr, 256, hipMemcpyDeviceToHost);
        uint32_t run = 0;
        for (int i = 0; i < nact; ++i) {
            run += h[i];
            const uint32_t ea = i + 1 < nact ? h[i + 1] : 0u;
            if (a[i] != ea) { printf("nact %d shl lane %d: got %u want %u\n", nact, i, a[i], ea); ++bad; }
            const uint32_t er = i > 0 ? h[i - 1] : 0u;
            if (r[i] != er) { printf("nact %d shr lane %d: got %u want %u\n", nact, i, r[i], er); ++bad; }
            if (s[i] != run) { printf("nact %d scan lane %d: got %u want %u\n", nact, i, s[i], run); ++bad; }
        }
    }
    printf(bad ? "DPP CHECK FAILED (%d)\n" : "DPP CHECK OK\n", bad);
    return bad != 0;
}
