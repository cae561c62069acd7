// The counter-based generator of the ISCO sampler steps (rls_isco.hip) and of K13's in-kernel partner draw (rls_tsp.hip): one
// definition, so that  ISCO_TSP.opt_2(sample, T)  under seed s draws what iteration 0 of the fused step draws under s.
#pragma once
#include "rls_common.h"

namespace rls {

// murmur3 finaliser as a counter-based generator (production draws; tests supply the reference's draws)
__device__ __forceinline__ uint32_t isco_mix(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
// draw(seed, env, a, b, stream) in three stages, so that a kernel pays each once: the env key (three rounds; wave-uniform where
// a wave owns one env -- scalar instructions), the position key (one round per counter a) and the draw (one round per
// (b, stream)).  The 32-bit multiplies of a round run at a quarter of the VALU rate: K13 recomputing all five rounds for each
// of its three draws was VALU-bound at 45 us where its bytes cost 24.
__device__ __forceinline__ uint32_t isco_env_key(uint64_t seed, uint64_t env) {
    uint32_t h = isco_mix((uint32_t)seed ^ 0x9E3779B9u);
    h = isco_mix(h ^ (uint32_t)(seed >> 32));
    return isco_mix(h ^ (uint32_t)env);
}
__device__ __forceinline__ uint32_t isco_pos_key(uint32_t env_key, uint64_t env, uint32_t a) {
    return isco_mix(env_key ^ (uint32_t)(env >> 32) ^ (a * 0x9E3779B1u));
}
__device__ __forceinline__ uint32_t isco_draw_at(uint32_t pos_key, uint32_t b, uint32_t stream) {
    return isco_mix(pos_key ^ (b * 0x85EBCA77u) ^ (stream * 0xC2B2AE3Du));
}
__device__ __forceinline__ uint32_t isco_draw(uint64_t seed, uint64_t env, uint32_t a, uint32_t b, uint32_t stream) {
    return isco_draw_at(isco_pos_key(isco_env_key(seed, env), env, a), b, stream);
}
// torch.rand-like uniform in [0, 1) with 24 bits
__device__ __forceinline__ float isco_unit(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

}  // namespace rls
