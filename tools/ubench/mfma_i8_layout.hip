// mfma_i8_layout.hip -- checks the operand and result maps of v_mfma_i32_16x16x64_i8 that the MFMA
// blur kernel relies on, with random signed bytes against a host product:
//   A fragment: lane l holds A[row l & 15][k = 16 (l >> 4) + b], b = byte 0..15 of its 4 dwords
//   B fragment: lane l holds B[k = 16 (l >> 4) + b][col l & 15]
//   C/D:        lane l, register v holds D[row 4 (l >> 4) + v][col l & 15]
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_i8_layout mfma_i8_layout.hip ; prints "layout ok" or the first mismatch
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k(const int8_t *A, const int8_t *B, int *D)   // A[16][64], B[64][16], D[16][16], row-major
{
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    union { v4i v; int8_t b[16]; } a, b;
    for (int i = 0; i < 16; i++) {
        a.b[i] = A[r * 64 + 16 * g + i];
        b.b[i] = B[(16 * g + i) * 16 + r];
    }
    v4i c = {1, 2, 3, 4};
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a.v, b.v, c, 0, 0, 0);
    for (int v = 0; v < 4; v++) D[(4 * g + v) * 16 + r] = c[v] - (v + 1);
}

int main()
{
    int8_t hA[16 * 64], hB[64 * 16];
    int hD[256], ref[256];
    srand(7);
    for (int i = 0; i < 1024; i++) { hA[i] = (int8_t)(rand() % 256 - 128); hB[i] = (int8_t)(rand() % 256 - 128); }
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            int s = 0;
            for (int kk = 0; kk < 64; kk++) s += (int)hA[i * 64 + kk] * (int)hB[kk * 16 + j];
            ref[i * 16 + j] = s;
        }
    int8_t *dA, *dB; int *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    if (hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 2; }
    for (int i = 0; i < 256; i++)
        if (hD[i] != ref[i]) { printf("mismatch at D[%d][%d]: %d, expected %d\n", i / 16, i % 16, hD[i], ref[i]); return 1; }
    printf("layout ok\n");
    return 0;
}
