/* placeholder until the LunarLander restatement lands */
#include <stdint.h>
#include <stdlib.h>
void* orc_lunar_alloc(int n) { (void)n; return NULL; }
void orc_lunar_reset_one(void* st, int i, uint64_t seed, uint64_t env, uint32_t episode, float* obs) { (void)st;(void)i;(void)seed;(void)env;(void)episode;(void)obs; abort(); }
void orc_lunar_step_one(void* st, int i, uint64_t seed, uint64_t env, int action, float* obs_next,
                        float* obs_term, float* rew, uint8_t* terminated, uint8_t* truncated,
                        int* done, double* ep_ret, int* ep_len) { (void)st;(void)i;(void)seed;(void)env;(void)action;(void)obs_next;(void)obs_term;(void)rew;(void)terminated;(void)truncated;(void)done;(void)ep_ret;(void)ep_len; abort(); }
