// api.hip -- version / error / workspace entry points of libcl3d.
#include "cl3d_common.h"

extern "C" int cl3d_abi_version(void) { return CL3D_ABI_VERSION; }

extern "C" const char *cl3d_last_error_string(void) { return cl3d::err_buf(); }

extern "C" size_t cl3d_workspace_bytes(int op, int B, int N, int M, int K, int C) {
  (void)B; (void)N; (void)M; (void)K; (void)C;
  switch (op) {
    // every op of ABI v1 keeps its scratch in LDS; the parameter exists so that the
    // multi-workgroup paths (large-N sort, cell lists) can ask for device scratch without an
    // ABI change.
    default:
      return 0;
  }
}
