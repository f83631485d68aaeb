// Fused score-network path -- placeholder until the fused kernels land: nothing is packed and
// fused_supported() is false, so every forward takes the generic path.
#include "fused.h"

namespace beso {

size_t fused_packed_bytes(const Layout&, int) { return 0; }
size_t fused_workspace_bytes(const Layout&, int, int, int) { return 0; }
int fused_pack(const Layout&, const float* const*, char*, int, hipStream_t) { return BESO_OK; }
bool fused_supported(const Layout&, const FwdArgs&, int) { return false; }
int forward_fused(const Layout&, const Workspace&, const char*, int, const FwdArgs&, char*, hipStream_t) {
    return BESO_ERR_UNSUPPORTED;
}

}  // namespace beso
