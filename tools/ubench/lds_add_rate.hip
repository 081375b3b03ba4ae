// lds_add_rate.hip -- what an LDS atomic add costs a CU, in the access patterns of the describe kernel:
// ds_add_u32 without return, 64 lanes onto 32 columns (lanes l and l+32 share a column), the row taken from
// per-lane data; compared with ds_write_b32, ds_add_u64 and with 64 distinct columns.
// Prints LDS instructions per microsecond and CU and the cycles per wave-instruction that implies at 2.3 GHz.
// build: hipcc --offload-arch=gfx950 -O2 -o lds_add_rate lds_add_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

enum { ADD32_SAMEROW, ADD32_ROWS, ADD32_ROWS_64COL, WRITE32_ROWS, ADD64_ROWS, ADD32_ROWS_16COL, ADD32_ROWS_SAMEDATA, NPAT };
const char *const kNames[NPAT] = {
    "ds_add_u32, 32 columns, one row per instruction",
    "ds_add_u32, 32 columns, row from lane data (the describe kernel)",
    "ds_add_u32, 64 columns, row from lane data",
    "ds_write_b32, 32 columns, row from lane data",
    "ds_add_u64, 16 column pairs x 4 lanes, row from lane data",
    "ds_add_u32, 16 columns, row from lane data",
    "ds_add_u32, 32 columns, row from lane data, address register reused for 3 adds at row +0/+8/+16",
};

// one wave per workgroup like k_describe; 6 KB of LDS per wave
template <int PAT>
__global__ __launch_bounds__(64) void k(const uint32_t *seed, uint32_t *out, int iters)
{
    __shared__ __align__(16) uint32_t lds[48 * 32];
    const int lane = threadIdx.x;
    for (int i = lane; i < 48 * 32; i += 64) lds[i] = 0;
    __syncthreads();
    uint32_t r = seed[(blockIdx.x * 64 + lane) & 1023];
    char *base = reinterpret_cast<char *>(lds);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            r += 0x9E3779B9u;
            const uint32_t row = (r >> 24) & 15u;          // 16 rows in use at a time (4 directions x a few circles)
            uint32_t addr;
            if (PAT == ADD32_SAMEROW) addr = (uint32_t)(u * 128) + (lane & 31) * 4;
            else if (PAT == ADD32_ROWS_64COL) addr = (row >> 1) * 256 + lane * 4;
            else if (PAT == ADD64_ROWS) addr = row * 128 + (lane & 15) * 8;
            else if (PAT == ADD32_ROWS_16COL) addr = row * 64 + (lane & 15) * 4;
            else addr = row * 128 + (lane & 31) * 4;
            uint32_t *p = reinterpret_cast<uint32_t *>(base + addr);
            if (PAT == WRITE32_ROWS) *reinterpret_cast<volatile uint32_t *>(p) = r;
            else if (PAT == ADD64_ROWS) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(p), (unsigned long long)(r & 0xff) | (unsigned long long)(r & 0xff00) << 24, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (PAT == ADD32_ROWS_SAMEDATA) {
                __hip_atomic_fetch_add(p, r & 0xffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(p + 8 * 32, r & 0xffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(p + 16 * 32, r & 0xffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            else __hip_atomic_fetch_add(p, r & 0xffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    uint32_t s = 0;
    for (int i = lane; i < 48 * 32; i += 64) s += lds[i];
    out[blockIdx.x * 64 + lane] = s;
}

template <int PAT>
void run(const uint32_t *d_seed, uint32_t *d_out, int waves_per_cu)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 256 * waves_per_cu;
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(64), 0, 0, d_seed, d_out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double per_wave = 16.0 * iters * (PAT == ADD32_ROWS_SAMEDATA ? 3 : 1);
    const double per_cu_us = per_wave * waves_per_cu / (best * 1e3);
    printf("%-100s %2d waves/CU: %7.3f ms, %6.1f instr/us/CU = %.2f cycles each at 2.3 GHz\n", kNames[PAT], waves_per_cu, best, per_cu_us, 2300.0 / per_cu_us);
}

int main()
{
    uint32_t h[1024];
    for (int i = 0; i < 1024; i++) h[i] = 0x9E3779B1u * (uint32_t)(i + 1);
    uint32_t *d_seed, *d_out;
    hipMalloc(&d_seed, sizeof h); hipMalloc(&d_out, 256 * 24 * 64 * 4);
    hipMemcpy(d_seed, h, sizeof h, hipMemcpyHostToDevice);
    for (int w : {8, 24}) {
        run<ADD32_SAMEROW>(d_seed, d_out, w);
        run<ADD32_ROWS>(d_seed, d_out, w);
        run<ADD32_ROWS_64COL>(d_seed, d_out, w);
        run<ADD32_ROWS_16COL>(d_seed, d_out, w);
        run<WRITE32_ROWS>(d_seed, d_out, w);
        run<ADD64_ROWS>(d_seed, d_out, w);
        run<ADD32_ROWS_SAMEDATA>(d_seed, d_out, w);
    }
    return 0;
}
