// als_cg.hip -- implicit-ALS half-epoch, conjugate-gradient solver (placeholder body).
#include "common.h"
struct lk_als_plan;
namespace lk {
int als_cg_half_epoch(const lk_als_plan *, const void *, int, const int32_t *, const float *,
                      int64_t, int, float *, int, const float *, int, const float *, int, char *,
                      float *, hipStream_t)
{
    set_error("CG solver not built yet");
    return LK_E_INVALID;
}
}  // namespace lk
