// v_sin_f32 / v_cos_f32 take revolutions and reduce to the fraction: is sin(1 + t) bit-identical to sin(t) for every t = m / 65536?
// (normal4 of the local search builds its angle as the float 1 + m / 65536 from the bits.)  Also the radius uniform as one fma.
// hipcc --offload-arch=gfx950 -O2 -o /tmp/sin_turn tools/ceilings/sin_turn.hip && /tmp/sin_turn
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(unsigned* bad) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= 65536) return;
    const float t = (float)m * (1.0f / 65536.0f);
    const float t1 = __builtin_bit_cast(float, (m << 7) | 0x3F800000u);
    if (__builtin_bit_cast(uint32_t, __builtin_amdgcn_sinf(t)) != __builtin_bit_cast(uint32_t, __builtin_amdgcn_sinf(t1))) {
        atomicAdd(&bad[0], 1u);
        printf("  sin differs at m = %u: %.9g (t) vs %.9g (1 + t)\n", m, __builtin_amdgcn_sinf(t), __builtin_amdgcn_sinf(t1));
    }
    if (__builtin_bit_cast(uint32_t, __builtin_amdgcn_cosf(t)) != __builtin_bit_cast(uint32_t, __builtin_amdgcn_cosf(t1))) atomicAdd(&bad[1], 1u);
    const float ua = ((float)m + 1.0f) * (1.0f / 65536.0f), ub = __builtin_fmaf((float)m, 1.0f / 65536.0f, 1.0f / 65536.0f);
    if (__builtin_bit_cast(uint32_t, ua) != __builtin_bit_cast(uint32_t, ub)) atomicAdd(&bad[2], 1u);
}
int main() {
    unsigned* d; unsigned h[3] = {0, 0, 0};
    hipMalloc(&d, sizeof h); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("65536 angles: sin mismatches %u, cos mismatches %u; radius uniform mismatches %u\n", h[0], h[1], h[2]);
    return (h[0] | h[1] | h[2]) ? 1 : 0;
}
