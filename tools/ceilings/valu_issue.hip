// Dev micro-benchmark: what the VALU issues per SIMD, instruction by instruction, for one, two and four waves per SIMD.
// Every wave runs ITERS x 64 independent register-only instructions (16 registers, no memory); one workgroup of 4 x wps waves per
// CU gives every SIMD of the chip `wps` waves.  Prints cycles per wave-instruction per SIMD, time x the NOMINAL 2.4 GHz.
// Measured on MI355X (round 4): one wave per SIMD 4.8-5.5 whatever the instruction; two or four waves: 2.4-2.6 for v_xor / v_and /
// v_bitop3 / v_add_u32 / v_mov / v_lshrrev / v_ashrrev / v_fma_f32 / v_mul_f32, 4.2-4.3 for v_lshlrev_b32 / v_lshl_or / v_and_or /
// v_or3 / v_bfi / v_bfe / v_perm / v_alignbit / v_bcnt / v_mul_lo / v_mul_u32_u24 / v_mad_u32_u24 / v_add3 / v_add_co / v_cvt /
// v_lshlrev_b64, 8.2 for v_log / v_sin / v_sqrt.  (The v_cndmask row is a dependent chain through vcc, not an issue rate.)
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_issue tools/ceilings/valu_issue.hip && /tmp/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int ITERS = 4096;

#define OPS(X)                                                                                  \
    X(0, "v_xor_b32", "v_xor_b32 %0, %2, %0", 2)                                                \
    X(1, "v_bitop3_b32", "v_bitop3_b32 %0, %0, %2, %3 bitop3:0xe8", 3)                          \
    X(2, "v_add_u32", "v_add_u32 %0, %2, %0", 2)                                                \
    X(3, "v_mul_lo_u32", "v_mul_lo_u32 %0, %2, %0", 2)                                          \
    X(4, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 3, %2", 2)                                     \
    X(5, "v_fma_f32", "v_fma_f32 %0, %2, %0, %0", 2)                                            \
    X(6, "v_bfe_i32", "v_bfe_i32 %0, %0, 3, 1", 1)                                              \
    X(7, "v_lshrrev_b32", "v_lshrrev_b32 %0, 3, %0", 1)                                         \
    X(8, "v_lshlrev_b32", "v_lshlrev_b32 %0, 1, %0", 1)                                         \
    X(9, "v_ashrrev_i32", "v_ashrrev_i32 %0, 31, %0", 1)                                        \
    X(10, "v_and_b32", "v_and_b32 %0, %2, %0", 2)                                               \
    X(11, "v_and_or_b32", "v_and_or_b32 %0, %0, %2, %3", 3)                                     \
    X(12, "v_add3_u32", "v_add3_u32 %0, %0, %2, %3", 3)                                         \
    X(13, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %2, vcc", 2)                                  \
    X(14, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %2, 7", 2)                                  \
    X(15, "v_perm_b32", "v_perm_b32 %0, %0, %2, %3", 3)                                         \
    X(16, "v_bcnt_u32_b32", "v_bcnt_u32_b32 %0, %2, %0", 2)                                     \
    X(17, "v_mov_b32", "v_mov_b32 %0, %2", 2)                                                   \
    X(18, "v_mul_u32_u24", "v_mul_u32_u24 %0, %2, %0", 2)                                       \
    X(19, "v_mad_u32_u24", "v_mad_u32_u24 %0, %2, %0, %0", 2)                                   \
    X(20, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0", 1)                                           \
    X(21, "v_log_f32", "v_log_f32 %0, %0", 1)                                                   \
    X(22, "v_sin_f32", "v_sin_f32 %0, %0", 1)                                                   \
    X(23, "v_sqrt_f32", "v_sqrt_f32 %0, %0", 1)                                                 \
    X(24, "v_mul_f32", "v_mul_f32 %0, %2, %0", 2)                                               \
    X(25, "v_lshlrev_b64", "v_lshlrev_b64 %1, 1, %1", 64)                                       \
    X(26, "v_cmp_gt_f32+cndmask", "v_cmp_gt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc", 2) \
    X(27, "v_add_co_u32", "v_add_co_u32 %0, vcc, %2, %0", 2)                                    \
    X(28, "v_bfi_b32", "v_bfi_b32 %0, %0, %2, %3", 3)                                           \
    X(29, "v_or3_b32", "v_or3_b32 %0, %0, %2, %3", 3)
constexpr int kNumOps = 30;

template <int OP>
__global__ __launch_bounds__(1024) void k_valu(uint32_t* out, uint32_t seed) {
    uint32_t r[16];
    uint64_t q[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = seed * (threadIdx.x + 1) + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = ((uint64_t)r[2 * i] << 32) | r[2 * i + 1];
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
#define X(ID, NAME, ASM, NSRC)                                                                                          \
    if constexpr (OP == ID && (ID == 13 || ID == 26 || ID == 27))                                                       \
        asm volatile(ASM : "+v"(r[i]), "+v"(q[i & 7]) : "v"(r[(i + 1) & 15]), "v"(r[(i + 2) & 15]) : "vcc");            \
    else if constexpr (OP == ID && ID == 25) asm volatile(ASM : "+v"(r[i]), "+v"(q[i & 7]));                            \
    else if constexpr (OP == ID) asm volatile(ASM : "+v"(r[i]) : "v"(r[(i + 1) & 15]), "v"(r[(i + 1) & 15]), "v"(r[(i + 2) & 15]));
                OPS(X)
#undef X
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s ^= r[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
    if (s == 0x12345678u) out[threadIdx.x] = s;
}

template <int OP>
static void launch_op(int op, dim3 grid, dim3 block, uint32_t* d) {
    if constexpr (OP < kNumOps) {
        if (op == OP) hipLaunchKernelGGL(k_valu<OP>, grid, block, 0, 0, d, 3u);
        else launch_op<OP + 1>(op, grid, block, d);
    }
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const char* names[kNumOps];
#define X(ID, NAME, ASM, NSRC) names[ID] = NAME;
    OPS(X)
#undef X
    printf("%d CUs, clock %d kHz; %d x 64 instructions per wave\n", cus, p.clockRate, ITERS);
    auto run = [&](int op, int wps) {   // wps waves per SIMD: one workgroup of 4 * wps waves per CU
        const dim3 grid(cus), block(256 * wps);
        auto launch = [&]() { launch_op<0>(op, grid, block, d); };
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = (double)ITERS * 64 * wps * 5;            // wave-instructions a SIMD issued
        printf("%-22s %d waves/SIMD: %.2f cycles per wave-instruction per SIMD at 2.4 GHz (%.3f ms)\n", names[op], wps,
               ms * 1e-3 * 2.4e9 / per_simd, ms / 5);
    };
    for (int op = 0; op < kNumOps; ++op)
        for (int wps : {1, 2, 4}) run(op, wps);
    return 0;
}
