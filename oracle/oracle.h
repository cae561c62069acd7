/* Prototypes of the C oracle (oracle/oracle.c) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see the header of oracle.c).
 * Included by oracle.c itself (so a drifted prototype is a compile error) and by tools/host_sanitize.cpp. */
#ifndef RLS_ORACLE_H
#define RLS_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int orc_num_threads(void);
void orc_maxcut_obj(const uint8_t* xs, int64_t B, int64_t N, const int32_t* eu, const int32_t* ev, int64_t E, int bidir, int64_t* out);
void orc_ppo_step(float* xs, int64_t B, int64_t N, const int64_t* action, const int32_t* eu, const int32_t* ev, int64_t E, int bidir,
                  float* last, float* reward, float* cur);
void orc_step_u8(uint8_t* xs, int64_t B, int64_t N, const int64_t* action, const int32_t* eu, const int32_t* ev, int64_t E, int bidir,
                 int64_t* last, int64_t* reward);
void orc_greedy_sweep(uint8_t* xs, int64_t B, int64_t N, const int32_t* eu, const int32_t* ev, int64_t E, int bidir, int64_t* vs);
void orc_node_cutdeg(const uint8_t* xs, int64_t B, int64_t N, const int32_t* erowptr, const int32_t* ev, int64_t* out);
void orc_tsp_tour_length(const float* dist, int64_t N, const int64_t* perm, int64_t B, float* out);
#ifdef __cplusplus
}
#endif
#endif
