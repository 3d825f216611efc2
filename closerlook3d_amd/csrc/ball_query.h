// ball_query.h -- internal interface between the exhaustive and the cell-grid ball query.
#pragma once
#include "cl3d_common.h"

namespace cl3d {

// exhaustive scan (ball_query.hip).  only_flagged == nullptr: every query; otherwise only queries j with
// only_flagged[b*M + j] != 0 are computed and written (workgroups without a flagged query exit at once).
int ball_query_exhaustive(const float *query_xyz, const float *support_xyz, const int *query_mask,
                          const int *support_mask, int B, int M, int N, float radius, int K, int *idx,
                          int *idx_mask, const int *only_flagged, hipStream_t st);

// cell-grid search (ball_query_cells.hip)
size_t ball_query_cells_workspace(int B, int N, int M);
bool ball_query_cells_applicable(int M, int N, int K);
int ball_query_cells(const float *query_xyz, const float *support_xyz, const int *query_mask,
                     const int *support_mask, int B, int M, int N, float radius, int K, int *idx,
                     int *idx_mask, void *ws, size_t ws_bytes, hipStream_t st);

// single-launch search with the cell-sorted cloud resident in LDS (ball_query_lds.hip): N, M <= 4096, no scratch
bool ball_query_tile_applicable(int M, int N, int K);
int ball_query_tile(const float *query_xyz, const float *support_xyz, const int *query_mask,
                    const int *support_mask, int B, int M, int N, float radius, int K, int *idx, int *idx_mask,
                    hipStream_t st);

// grid_subsample.hip
size_t grid_subsampling_workspace(int B, int N);
// csr.hip
size_t inverse_index_workspace(int B, int N, int MK);
size_t dataset_grid_workspace(int n);
// sphere_crop.hip: sort keys / values (x2) and rocPRIM temporary storage for P scene points (or num_points slots)
size_t sphere_crop_workspace(int P);
// mfma_gemm.hip: scratch (K-slice partial tiles) of the three products of a per-point contraction rows_in -> rows_out
// over nb clouds of n points; merge: the PointWiseMLP form, whose weight gradient always goes through the reduce
size_t gemm_family_workspace(int nb, int n, int rows_out, int rows_in, bool merge);

}  // namespace cl3d
