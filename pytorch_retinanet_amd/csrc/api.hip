// Version / status strings of libretinanet_hip.so.
#include "rn_common.hpp"

RN_API int rn_version(void) { return RN_ABI_VERSION; }

RN_API const char *rn_status_string(int status)
{
    switch (status) {
        case RN_OK: return "ok";
        case RN_EINVAL: return "invalid argument (null pointer, non-positive size or bad enum)";
        case RN_EALIGN: return "pointer alignment requirement not met";
        case RN_EWORKSPACE: return "workspace too small";
        case RN_EUNSUPPORTED: return "shape outside the supported range";
        case RN_ETHRESH: return "match threshold must be greater than background threshold";
        default: break;
    }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "unknown status";
}
