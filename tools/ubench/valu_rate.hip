// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction per SIMD of the integer
// ops the kernels are built from, at 1/2/4 waves per SIMD.  Each op is one inline-asm instruction
// on 16 independent accumulators, so nothing is folded away.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ACC 16
#define ITERS 2048
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return; } } while (0)

#define OPS(X) \
    X(0, "v_dot4_u32_u8 %0, %1, %2, %0") \
    X(1, "v_add_u32 %0, %1, %0") \
    X(2, "v_alignbyte_b32 %0, %0, %1, 1") \
    X(3, "v_mad_u32_u24 %0, %1, %2, %0") \
    X(4, "v_lshl_add_u32 %0, %0, 8, %1") \
    X(5, "v_and_or_b32 %0, %0, %1, %2") \
    X(6, "v_fma_f32 %0, %1, %2, %0") \
    X(7, "v_pk_add_u16 %0, %0, %1") \
    X(8, "v_pk_mad_u16 %0, %1, %2, %0") \
    X(9, "v_perm_b32 %0, %0, %1, %2") \
    X(10, "v_add3_u32 %0, %0, %1, %2") \
    X(11, "v_dot2_u32_u16 %0, %1, %2, %0") \
    X(12, "v_mul_u32_u24 %0, %0, %1") \
    X(13, "v_lshrrev_b32 %0, 1, %0") \
    X(14, "v_bfe_u32 %0, %0, 1, 30") \
    X(15, "v_mov_b32 %0, %1") \
    X(16, "v_pk_mul_lo_u16 %0, %0, %1") \
    X(17, "v_max_u32 %0, %0, %1") \
    X(18, "v_cndmask_b32 %0, %0, %1, vcc") \
    X(19, "v_sad_u8 %0, %1, %2, %0") \
    X(20, "v_dot8_u32_u4 %0, %1, %2, %0") \
    X(21, "v_xor_b32 %0, %0, %1") \
    X(22, "v_pk_fma_f32 %0, %1, %2, %0") \
    X(23, "v_dot4c_i32_i8 %0, %1, %2") \
    X(24, "v_dot4_i32_i8 %0, %1, %2, %0") \
    X(25, "v_fmac_f32 %0, %1, %2") \
    X(26, "v_and_b32 %0, %0, %1") \
    X(27, "v_sub_u32 %0, %0, %1") \
    X(28, "v_lshlrev_b32 %0, 1, %0") \
    X(29, "v_or_b32 %0, %0, %1") \
    X(30, "v_dot2c_i32_i16 %0, %1, %2") \
    X(31, "v_mul_lo_u32 %0, %0, %1") \
    X(32, "v_min_u32 %0, %0, %1") \
    X(33, "v_ashrrev_i32 %0, 1, %0") \
    X(34, "v_sad_u32 %0, %1, %2, %0") \
    X(35, "v_dot4_u32_u8 %0, %1, %2, %0\n\tv_add_u32 %0, %1, %0") \
    X(36, "v_dot4_u32_u8 %0, %1, %2, %0\n\tv_dot4_u32_u8 %0, %1, %2, %0\n\tv_dot4_u32_u8 %0, %1, %2, %0\n\tv_add_u32 %0, %1, %0") \
    X(37, "v_add_u32 %0, %1, %0\n\tv_lshlrev_b32 %0, 1, %0") \
    X(38, "; v_lshl_add_u64 (two 32-bit adds in one)") \
    X(39, "v_pk_maximum3_f16 %0, %0, %1, %2") \
    X(40, "v_pk_minimum3_f16 %0, %0, %1, %2") \
    X(41, "v_pk_max_i16 %0, %0, %1") \
    X(42, "v_pk_sub_i16 %0, %0, %1") \
    X(43, "v_pk_add_f16 %0, %0, %1") \
    X(44, "v_max3_i32 %0, %0, %1, %2") \
    X(45, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1") \
    X(46, "v_cmp_gt_f16_sdwa vcc, %0, %1 src0_sel:WORD_1 src1_sel:WORD_1") \
    X(47, "v_cmp_gt_i32 vcc, %0, %1") \
    X(48, "v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,1,0,0]") \
    X(49, "v_dot2_i32_i16 %0, %1, %2, %0") \
    X(50, "v_alignbit_b32 %0, %0, %1, 16") \
    X(51, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf") \
    X(52, "v_pk_lshrrev_b16 %0, 8, %0") \
    X(53, "v_sqrt_f32 %0, %0") \
    X(54, "v_cvt_f32_i32 %0, %0") \
    X(55, "v_pk_max_f16 %0, %0, %1") \
    X(56, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8")

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a[ACC], x = seed + threadIdx.x, y = seed * 3 + threadIdx.x;
    unsigned long long a2[ACC], x2 = x, y2 = y;
#pragma unroll
    for (int i = 0; i < ACC; i++) { a[i] = i + threadIdx.x; a2[i] = a[i]; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ACC; i++) {
#define X(n, s) if (OP == n && n != 22 && n != 38) asm volatile(s : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
            OPS(X)
#undef X
            if (OP == 22) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2[i]) : "v"(x2), "v"(y2));
            if (OP == 38) asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(a2[i]) : "v"(x2));
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < ACC; i++) s ^= a[i] ^ (uint32_t)a2[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, uint32_t *d)
{
    printf("%-36s", name);
    for (int wps = 1; wps <= 4; wps *= 2) {          // 256 blocks x 256 threads = 1 wave per SIMD
        int blocks = 256 * wps;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        k<OP><<<blocks, 256>>>(d, 1);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) k<OP><<<blocks, 256>>>(d, r);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 5;
        printf("  %dw/SIMD %.2f", wps, ms * 1e-3 * 2.4e9 / ((double)ITERS * ACC * wps));
    }
    printf("   (cycles @2.4 GHz per wave-instr per SIMD)\n");
}

int main()
{
    uint32_t *d;
    if (hipMalloc(&d, 256 * 4 * 256 * 4) != hipSuccess) return 1;
#define X(n, s) run<n>(s, d);
    OPS(X)
#undef X
    return 0;
}
