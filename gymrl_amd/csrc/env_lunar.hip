// placeholder until the LunarLander solver lands
#include "env_common.hpp"
namespace gymrl {
size_t lunar_state_bytes(int) { return 0; }
int lunar_reset(void*, int, uint64_t, int64_t, float*, hipStream_t) { return -38; }
int lunar_step(void*, int, uint64_t, int64_t, const int32_t*, float*, float*, float*, uint8_t*,
               uint8_t*, uint8_t*, float*, int32_t*, double*, hipStream_t) { return -38; }
}
