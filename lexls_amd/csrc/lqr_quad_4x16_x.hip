#include "lqr_quad_impl.h"
LEXLS_QUAD_INSTANCE(launch_quad_4x16_x, 4, 16, false, 0)
namespace lexls
{
    /// dynamic LDS one wavefront (four problems) of the four-per-wavefront kernel asks for; 0 = the shape is not served
    size_t quad_lds_bytes(uint32_t slots, uint32_t md, uint32_t nVar, uint32_t nObj)
    {
        if (nObj > (uint32_t)kQuadMaxObj || nVar + 1 > 16u * slots || nVar > 63u) return 0;
        const size_t g = slots == 3 ? quad_group_bytes<3>(nVar, nObj, md) : quad_group_bytes<4>(nVar, nObj, md);
        return 4 * g;
    }
} // namespace lexls
