// api.hip -- version / error / workspace entry points of libcl3d.
#include "ball_query.h"
#include "fused_common.h"

extern "C" int cl3d_abi_version(void) { return CL3D_ABI_VERSION; }

extern "C" int cl3d_d2_form(void) { return CL3D_D2_FORM; }

extern "C" int cl3d_fused_supported(int op, int K, int C) {
  switch (op) {
    case CL3D_OP_POSPOOL: return cl3d::fused_reduce_supported(0, K, C) && cl3d::fused_reduce_supported(1, K, C);
    case CL3D_OP_ADAPTIVE_WEIGHT: return cl3d::fused_reduce_supported(2, K, C);
    case CL3D_OP_PSEUDO_GRID: return cl3d::fused_reduce_supported(3, K, C);
    case CL3D_OP_POINTWISE_MLP: return cl3d::pwmlp_supported(K, C);
    case CL3D_OP_MAX_POOL: return cl3d::maxpool_supported(K, C);
    default: return 0;
  }
}

extern "C" const char *cl3d_last_error_string(void) { return cl3d::err_buf(); }

extern "C" size_t cl3d_workspace_bytes(int op, int B, int N, int M, int K, int C) {
  switch (op) {
    case CL3D_OP_BALL_QUERY:  // cell grid: sorted support copy, cell starts, query order, task table, flags
      return cl3d::ball_query_cells_applicable(M, N, K) ? cl3d::ball_query_cells_workspace(B, N, M) : 0;
    case CL3D_OP_GRID_SUBSAMPLING:  // clouds beyond the in-LDS sort: 64-bit keys (x2) + rocPRIM temporary storage
      return cl3d::grid_subsampling_workspace(B, N);
    case CL3D_OP_INVERSE_INDEX:  // sort keys/values + rocPRIM temporary storage; M*K slots per cloud
      return cl3d::inverse_index_workspace(B, N, M * K);
    case CL3D_OP_DATASET_GRID:  // keys and order (x2), heads, ranks, rocPRIM temporary storage; one cloud of N points
      return cl3d::dataset_grid_workspace(N);
    case CL3D_OP_SPHERE_CROP:  // sort keys and order (x2), rocPRIM temporary storage; N scene points or sample slots
      return cl3d::sphere_crop_workspace(N);
    case CL3D_OP_POINT_GEMM:  // PointWiseMLP per-point contraction: M carries Co; K-slice partials of any of its three products
      return cl3d::gemm_family_workspace(B, N, 2 * M, C, true);
    case CL3D_OP_CONV1X1:  // 1x1 Conv1d C -> M over B clouds of N points: K-slice partials of any of its three products
      return cl3d::gemm_family_workspace(B, N, M, C, false);
    default:  // every other op of ABI v1 keeps its scratch in LDS
      return 0;
  }
}
