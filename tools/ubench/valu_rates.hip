// Integer VALU issue rates on gfx950: cycles per wave64 instruction per SIMD for the ops the kernels use,
// at 1, 2, 4, 8 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_ITER 2000
#define UNROLL 32

template <int OP>
__global__ void k(unsigned* out, unsigned seed)
{
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * (i + 1);
    unsigned s1 = seed | 1, s2 = seed * 3 + 7;
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            unsigned& x = a[u & 7];
            const unsigned y = a[(u + 3) & 7];          // cross-element dependency: nothing folds
            if (OP == 0) x = x + y;                                               // v_add_u32
            if (OP == 1) x = (unsigned)max(max((int)x, (int)y), (int)s1);         // v_max3_i32
            if (OP == 2) x = (x << 3) | y;                                        // v_lshl_or_b32
            if (OP == 3) x = __builtin_amdgcn_ubfe(x ^ y, 5, 7);                  // v_xor + v_bfe_u32
            if (OP == 4) x = __builtin_amdgcn_perm(x, y, s2);                     // v_perm_b32
            if (OP == 5) x = __builtin_amdgcn_udot4(x, y, s2, false);             // v_dot4_u32_u8
            if (OP == 6) x = __popc(x) + y;                                       // v_bcnt_u32_b32 (popc + add in one)
            if (OP == 7) x = __builtin_amdgcn_alignbit(x, y, 7);                  // v_alignbit_b32
            if (OP == 8) x = __builtin_amdgcn_sad_u8(x, y, s2);                   // v_sad_u8
            if (OP == 9) x = (x & y) | s2;                                        // v_and_or_b32
            if (OP == 10) x = x * y;                                              // v_mul_lo_u32
            if (OP == 11) x = (unsigned)max((int)x, (int)y);                      // v_max_i32
            if (OP == 12) x = x ^ y;                                              // v_xor_b32
            if (OP == 13) x = ((x >> 5) & 0x7Fu) ^ y;                             // shift + and + xor (bfe alternative)
            if (OP == 14) x = (x << 12) + y;                                      // v_lshl_add_u32
            if (OP == 15) x = x | y | s2;                                         // v_or3_b32
            if (OP == 16) x = (unsigned)((int)x - (int)y);                        // v_sub_u32
            if (OP == 17) x = __builtin_amdgcn_ubfe(x, 5, 7) ^ y;                 // v_bfe_u32 + v_xor
            if (OP == 18) x = (x << 3) ^ y;                                       // v_lshlrev + v_xor
            if (OP == 19) x = (x >> 3) ^ y;                                       // v_lshrrev + v_xor
            if (OP == 20) x = ((int)x < (int)y) ? (x ^ s1) : (y ^ s2);            // v_cmp + 2 xor + v_cndmask
            if (OP == 21) { typedef short s2v __attribute__((ext_vector_type(2))); s2v p = __builtin_bit_cast(s2v, x), q = __builtin_bit_cast(s2v, y);
                            s2v r = __builtin_elementwise_max(p, q); x = __builtin_bit_cast(unsigned, r); }      // v_pk_max_i16
            if (OP == 22) { typedef unsigned short u2v __attribute__((ext_vector_type(2))); u2v p = __builtin_bit_cast(u2v, x), q = __builtin_bit_cast(u2v, y);
                            u2v r = p + q; x = __builtin_bit_cast(unsigned, r); }                               // v_pk_add_u16
            if (OP == 23) x = x + y + s1;                                         // v_add3_u32
            if (OP == 24) x = (unsigned)min((int)x, (int)y);                      // v_min_i32
            if (OP == 25) x = (x & 0xFFFFFFu) * 5u + y;                           // v_and + v_mad_u32_u24
            if (OP == 26) x = max(x, y);                                          // v_max_u32
            if (OP == 27) x = __shfl_xor((int)x, 1) ^ y;                          // ds_swizzle/bpermute + xor
            if (OP == 28) x = __builtin_amdgcn_mov_dpp(x, 0x111, 0xF, 0xF, true) ^ y;   // v_mov_b32 dpp row_shr:1 + xor
        }
    }
    unsigned r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
void run(const char* name, unsigned* d_out)
{
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    for (int wps : {8}) {
        const int blocks = 256 * 1;                 // one 256*wps-thread... use wps blocks per CU of 256 threads = wps waves per SIMD
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(blocks * wps), dim3(256), 0, 0, d_out, 1u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks * wps), dim3(256), 0, 0, d_out, 1u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)N_ITER * UNROLL * wps;            // wave-instructions issued on one SIMD
        const double cyc = ms * 1e-3 * clk_khz * 1e3 / instr_per_simd;
        printf("%-16s waves/SIMD %d : %.3f ms  %.2f cycles per wave-instruction per SIMD (at %d MHz nominal)\n", name, wps, ms, cyc, clk_khz / 1000);
    }
}

int main()
{
    unsigned* d_out; hipMalloc(&d_out, sizeof(unsigned) * 256 * 8 * 256);
    run<0>("v_add_u32", d_out); run<1>("v_max3_i32", d_out); run<2>("v_lshl_or_b32", d_out); run<3>("v_xor+v_bfe", d_out);
    run<4>("v_perm_b32", d_out); run<5>("v_dot4_u32_u8", d_out); run<6>("v_bcnt+add", d_out); run<7>("v_alignbit_b32", d_out);
    run<8>("v_sad_u8", d_out); run<9>("v_and_or_b32", d_out); run<10>("v_mul_lo_u32", d_out); run<11>("v_max_i32", d_out);
    run<12>("v_xor_b32", d_out); run<13>("shr+and+xor", d_out); run<14>("v_lshl_add_u32", d_out); run<15>("v_or3_b32", d_out); run<16>("v_sub_u32", d_out);
    run<17>("v_bfe+v_xor", d_out); run<18>("v_lshl+v_xor", d_out); run<19>("v_lshr+v_xor", d_out); run<20>("cmp+2xor+cndmask", d_out);
    run<21>("v_pk_max_i16", d_out); run<22>("v_pk_add_u16", d_out); run<23>("v_add3_u32", d_out); run<24>("v_min_i32", d_out);
    run<25>("and+mad_u24", d_out); run<26>("v_max_u32", d_out); run<27>("shfl_xor+xor", d_out); run<28>("dpp_mov+xor", d_out);
    return 0;
}
