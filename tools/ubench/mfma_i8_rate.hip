// mfma_i8_rate.hip -- sustained rate of v_mfma_i32_16x16x64_i8 with the operand pattern of the blur
// kernel (signed pixel bytes, small non-negative coefficients), whole chip, 1 and 2 waves per SIMD.
// Prints MFMA/s per SIMD and the clock that implies at 16 cycles per MFMA.  The second kernel is the
// K = 32 form (v_mfma_i32_16x16x32_i8, 8 operand bytes per lane): VERDICT r2 item 3a asks what it costs.
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_i8_rate mfma_i8_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int v4i __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void k(const int *seed, int *out, int iters)
{
    v4i a, b, acc[NACC];
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    for (int i = 0; i < 4; i++) { a[i] = (seed[(t + i) & 1023] & 0x7f7f7f7f); b[i] = seed[(t * 7 + i) & 1023]; }
    for (int k2 = 0; k2 < NACC; k2++) acc[k2] = v4i{k2, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k2 = 0; k2 < NACC; k2++) acc[k2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[k2], 0, 0, 0);
        b[0] ^= it;                                  // operands change, as in the real kernel
    }
    int s = 0;
    for (int k2 = 0; k2 < NACC; k2++) s += acc[k2][0] + acc[k2][1] + acc[k2][2] + acc[k2][3];
    out[t] = s;
}

template <int NACC>
__global__ __launch_bounds__(512) void k32(const int *seed, int *out, int iters)
{
    long a, b;
    v4i acc[NACC];
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    a = ((long)(seed[t & 1023] & 0x7f7f7f7f) << 32) | (unsigned)(seed[(t + 1) & 1023] & 0x7f7f7f7f);
    b = ((long)seed[(t * 7) & 1023] << 32) | (unsigned)seed[(t * 7 + 1) & 1023];
    for (int k2 = 0; k2 < NACC; k2++) acc[k2] = v4i{k2, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k2 = 0; k2 < NACC; k2++) acc[k2] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, acc[k2], 0, 0, 0);
        b ^= it;
    }
    int s = 0;
    for (int k2 = 0; k2 < NACC; k2++) s += acc[k2][0] + acc[k2][1] + acc[k2][2] + acc[k2][3];
    out[t] = s;
}

int main()
{
    int h[1024];
    for (int i = 0; i < 1024; i++) h[i] = (int)(0x9E3779B1u * (unsigned)(i + 1));
    int *d_seed, *d_out;
    hipMalloc(&d_seed, sizeof h); hipMalloc(&d_out, 512 * 1024 * 4);
    hipMemcpy(d_seed, h, sizeof h, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int NACC = 16, iters = 20000;
    for (int form = 0; form < 2; form++)
    for (int waves_per_simd = 1; waves_per_simd <= 2; waves_per_simd++) {
        const int threads = 256 * waves_per_simd, blocks = 256;      // one block per CU
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0, 0);
            if (form == 0) hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d_seed, d_out, iters);
            else hipLaunchKernelGGL(k32<NACC>, dim3(blocks), dim3(threads), 0, 0, d_seed, d_out, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double per_simd = (double)NACC * iters * waves_per_simd / (ms * 1e-3);
            printf("16x16x%d, %d wave(s)/SIMD: %.3f ms, %.1f M MFMA/s per SIMD -> %.2f GHz at 16 cycles, %.2f PMAC/s chip\n",
                   form ? 32 : 64, waves_per_simd, ms, per_simd / 1e6, per_simd * 16 / 1e9,
                   per_simd * 1024 * (form ? 8192 : 16384) / 1e15);
        }
    }
    return 0;
}
