/*
 * cl3d.h -- C ABI of the MI355X-native local-aggregation engine (libcl3d.so).
 *
 * This is the drop-in boundary for the hot path of zeliu98/CloserLook3D
 * (SURVEY.md 8(b)).  Every entry point replaces one function of the reference's
 * pybind module `pt_custom_ops._ext`
 * (pytorch/ops/pt_custom_ops/_ext_src/src/bindings.cpp:8-14) or one fused step of
 * pytorch/ops/pt_custom_ops/pt_utils.py / models/local_aggregation_operators.py.
 *
 * Conventions (all entry points):
 *   - plain device pointers + explicit sizes, no framework types;
 *   - tensors are dense/contiguous, float32 or int32, laid out exactly as the
 *     reference lays them out: xyz [B,N,3] (AoS), masks [B,N] int32 (valid points
 *     first), features channel-major [B,C,N], neighbour indices [B,M,K] int32;
 *   - inputs are borrowed and never written; outputs are fully overwritten (no
 *     pre-zeroing required, unlike the reference's torch::zeros + kernel);
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) and the call returns without synchronising;
 *   - return value: CL3D_OK (0) or a negative CL3D_E_* code; the reference's
 *     `exit(-1)` on launch failure (cuda_utils.h:35-44) is NOT reproduced.
 *     cl3d_last_error_string() gives the message for the calling thread;
 *   - `ws`/`ws_bytes`: caller-provided device scratch, size from
 *     cl3d_workspace_bytes(); contents are undefined before and after a call.
 */
#ifndef CL3D_H_
#define CL3D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: round 3 -- cl3d_pwmlp_bwd_rows / cl3d_pwmlp_bwd_support changed their argument lists (query table and
 * point-major dz rows); the round-2 changes to bn_relu_stats / fused_reduce / pwmlp_* had been made under version 1
 * 3: round 4 -- cl3d_pwmlp_support_summary / cl3d_pwmlp_bwd_support_sum are gone (cl3d_pwmlp_bwd_support is the one
 * support-major pass again, same argument list as in version 2); cl3d_pwmlp_train_forward / _backward added (one call
 * per pass, csrc/pass.hip); cl3d_sphere_crop_assemble takes the capacity of the index list it is handed */
#define CL3D_ABI_VERSION 5

#define CL3D_OK 0
#define CL3D_E_INVALID (-1)     /* bad argument (null pointer, negative size, ...) */
#define CL3D_E_LAUNCH (-2)      /* HIP reported an error at launch */
#define CL3D_E_WORKSPACE (-3)   /* ws_bytes smaller than cl3d_workspace_bytes() */
#define CL3D_E_UNSUPPORTED (-4) /* size outside what this build supports */

typedef void *cl3d_stream_t; /* hipStream_t */

/* operation ids for cl3d_workspace_bytes */
#define CL3D_OP_BALL_QUERY 1
#define CL3D_OP_GROUP_POINTS 2
#define CL3D_OP_GROUP_POINTS_GRAD 3
#define CL3D_OP_GRID_SUBSAMPLING 4
#define CL3D_OP_NEAREST_QUERY 5
#define CL3D_OP_QUERY_AND_GROUP 6
#define CL3D_OP_POSPOOL 7
#define CL3D_OP_ADAPTIVE_WEIGHT 8
#define CL3D_OP_PSEUDO_GRID 9
#define CL3D_OP_POINTWISE_MLP 10
#define CL3D_OP_INVERSE_INDEX 11 /* cl3d_build_inverse_index: pass M*K slots as (M, K) */
#define CL3D_OP_DATASET_GRID 12  /* cl3d_dataset_grid_subsampling: N = points of the cloud (B, M, K, C unused) */
#define CL3D_OP_MAX_POOL 13      /* cl3d_maxpool_fwd/bwd (only meaningful for cl3d_fused_supported) */
#define CL3D_OP_POINT_GEMM 14    /* cl3d_pwmlp_point_gemm_bwd_weight: (B, N, M = Co, K unused, C) */
#define CL3D_OP_CONV1X1 15       /* cl3d_conv1x1_bwd_weight: (B, N, M = Cout, K unused, C = Cin) */
#define CL3D_OP_SPHERE_CROP 16   /* cl3d_sphere_crop_query: N = scene points; _assemble: N = num_points */

int cl3d_abi_version(void);
const char *cl3d_last_error_string(void);
/* operation order of the squared distance this build compares against radius^2 (build-time CL3D_D2_FORM, see
 * closerlook3d_amd/build.py): 0 = fadd(fma(dy,dy,dx*dx), dz*dz) -- hipcc's contraction of
 * masked_ordered_ball_query_gpu.cu:56-57 on gfx950 (default); 1 = no contraction; 2 = full fma chain. */
int cl3d_d2_form(void);
/* 1 when the fused kernels of operator `op` (CL3D_OP_POSPOOL / _ADAPTIVE_WEIGHT / _PSEUDO_GRID / _POINTWISE_MLP /
 * _MAX_POOL) take this nsample K and channel count C (output channels for the PointWiseMLP): their per-block slot
 * tile lives in LDS and grows with K.  0 -> the caller runs the grouped dataflow on the five native ops instead
 * (the fused entry points would return CL3D_E_UNSUPPORTED). */
int cl3d_fused_supported(int op, int K, int C);
/* bytes of device scratch the op needs for these sizes (0 if none). Unused dims: pass 0. */
size_t cl3d_workspace_bytes(int op, int B, int N, int M, int K, int C);

/* ---- the five legacy native ops ------------------------------------------------------ */

/* replaces masked_ordered_ball_query (masked_ordered_ball_query.cpp:13-59 +
 * masked_ordered_ball_query_gpu.cu:11-96). idx, idx_mask: [B,M,nsample] int32. */
int cl3d_masked_ordered_ball_query(const float *query_xyz, const float *support_xyz,
                                   const int32_t *query_mask, const int32_t *support_mask, int B,
                                   int M, int N, float radius, int nsample, int32_t *idx,
                                   int32_t *idx_mask, void *ws, size_t ws_bytes,
                                   cl3d_stream_t stream);

/* the same op on a named implementation: path 0 = the library's choice (identical to the call above), 1 = tile (the cloud
 * resident in one CU's LDS; N, M <= 4096), 2 = cells (cell grid through HBM scratch), 3 = exhaustive scan.  Same bits
 * from every path; CL3D_E_UNSUPPORTED when the path does not take the sizes.  cl3d_ball_query_paths: bit p set = path p
 * applies.  Which of tile / cells is faster depends on the point density inside the radius (the reference's ">3K
 * candidates" regime, masked_ordered_ball_query_gpu.cu:58-75) -- a property of the data: a caller with a static geometry
 * times the applicable paths once and keeps the winner (closerlook3d_amd/pt_utils.py does). */
int cl3d_masked_ordered_ball_query_path(int path, const float *query_xyz, const float *support_xyz,
                                        const int32_t *query_mask, const int32_t *support_mask, int B, int M,
                                        int N, float radius, int nsample, int32_t *idx, int32_t *idx_mask,
                                        void *ws, size_t ws_bytes, cl3d_stream_t stream);
int cl3d_ball_query_paths(int M, int N, int nsample);

/* replaces group_points (group_points.cpp:17-40 + group_points_gpu.cu:13-33).
 * points [B,C,N], idx [B,M,K] -> out [B,C,M,K]. */
int cl3d_group_points(const float *points, const int32_t *idx, int B, int C, int N, int M, int K,
                      float *out, cl3d_stream_t stream);

/* replaces group_points_grad (group_points.cpp:42-65 + group_points_gpu.cu:48-69).
 * grad_out [B,C,M,K], idx [B,M,K] -> grad_points [B,C,N] (sum over all (j,k) with idx==i,
 * accumulated in double and rounded once: order-independent up to the final rounding, where the
 * reference's float atomicAdd depends on scheduling). */
int cl3d_group_points_grad(const float *grad_out, const int32_t *idx, int B, int C, int N, int M,
                           int K, float *grad_points, void *ws, size_t ws_bytes,
                           cl3d_stream_t stream);

/* replaces masked_grid_subsampling (masked_grid_subsampling.cpp:13-44 +
 * masked_grid_subsampling_gpu.cu:11-153). xyz [B,N,3], mask [B,N] ->
 * sub_xyz [B,m,3], sub_mask [B,m]. */
int cl3d_masked_grid_subsampling(const float *xyz, const int32_t *mask, int B, int N, int m,
                                 float sampleDl, float *sub_xyz, int32_t *sub_mask, void *ws,
                                 size_t ws_bytes, cl3d_stream_t stream);

/* replaces masked_nearest_query (masked_nearest_query.cpp:12-47 +
 * masked_nearest_query_gpu.cu:8-62). idx, idx_mask: [B,M,1] int32. */
int cl3d_masked_nearest_query(const float *query_xyz, const float *support_xyz,
                              const int32_t *query_mask, const int32_t *support_mask, int B, int M,
                              int N, int32_t *idx, int32_t *idx_mask, cl3d_stream_t stream);

/* ---- fused grouping: one launch for what MaskedQueryAndGroup.forward does after the ball
 * query (pt_utils.py:125-132): rel[b,a,j,k] = (support_xyz[b,idx,a] - query_xyz[b,j,a]) * inv_scale
 * (inv_scale = 1/radius when normalize_xyz, else 1; the reference divides, see note in
 * DESIGN.md) and, when features != NULL, grouped[b,c,j,k] = features[b,c,idx].
 * rel [B,3,M,K]; grouped [B,C,M,K] (may be NULL together with features). */
int cl3d_group_xyz_features(const float *query_xyz, const float *support_xyz,
                            const float *features, const int32_t *idx, int B, int C, int N, int M,
                            int K, float radius, int normalize_xyz, float *rel, float *grouped,
                            cl3d_stream_t stream);

/* Dataset-side grid subsampling of ONE cloud (datasets/data_utils.py:12-30 -> ops/cpp_wrappers/cpp_subsampling/
 * grid_subsampling/grid_subsampling.cpp:5-106): voxel barycentres [*,3], feature means [*,fdim] (features
 * nullable with fdim = 0) and majority labels [*,ldim] (labels nullable with ldim = 0) of points [N,3]; outputs
 * need room for N rows, *count receives the number of voxels.  Same float arithmetic as the reference (sums in
 * original point order); voxels come out in ascending (iz,iy,ix) order and label ties go to the smallest label
 * (the reference's order and tie-break are those of an unordered_map walk, i.e. unspecified).
 * ws: cl3d_workspace_bytes(CL3D_OP_DATASET_GRID, 1, N, 0, 0, 0). */
int cl3d_dataset_grid_subsampling(const float *points, const float *features, const int32_t *labels, int N,
                                  int fdim, int ldim, float sampleDl, float *sub_points, float *sub_features,
                                  int32_t *sub_labels, int32_t *count, void *ws, size_t ws_bytes,
                                  cl3d_stream_t stream);

/* S3DIS sphere crop of a scene resident in HBM (datasets/S3DIS.py:296-314: KDTree.query_radius(pick, r,
 * return_distance=True, sort_results=True), the num_points nearest, shuffled, padded by re-drawn valid points).
 * query: points [P,3] float64 (the tree's own copies), pick: 3 doubles ON THE HOST; *count (device) = the number of
 * points with rdist = (dx*dx + dy*dy) + dz*dz <= radius^2; sorted_idx [cap] receives them ordered by (float64 distance
 * sqrt(rdist), index) -- the tree's sorted in-radius list -- when *count <= cap (1 <= cap <= P: the host-known size of
 * the sort; cap = P always suffices, a smaller bound sorts less).  *count > cap: repeat with a larger cap.
 * assemble: one sample of num_points slots from (sorted_idx [cap], count) -- *count > cap (the list was not written):
 * nothing of it is read, every slot gets point 0 with mask 0, the caller repeats the query -- otherwise
 * m = min(*count, num_points) nearest points in
 * the order of their draws u_shuffle [num_points] (uniform [0,1), caller's RNG), then slots >= m re-draw one of them
 * by u_redraw; out_inds int64, out_mask, out_points = float32(point - pick), out_height = float32(z).
 * ws: cl3d_workspace_bytes(CL3D_OP_SPHERE_CROP, 1, P, 0, 0, 0) (query, any cap; also enough for assemble). */
int cl3d_sphere_crop_query(const double *points, int P, const double *pick, double radius, int cap,
                           int32_t *sorted_idx, int32_t *count, void *ws, size_t ws_bytes, cl3d_stream_t stream);
int cl3d_sphere_crop_assemble(const double *points, const int32_t *sorted_idx, const int32_t *count, int cap,
                              int num_points, const double *pick, const float *u_shuffle, const float *u_redraw, float *out_points,
                              int32_t *out_mask, int64_t *out_inds, float *out_height, void *ws, size_t ws_bytes,
                              cl3d_stream_t stream);

/* BatchNorm1d + ReLU on channel-major x [B,C,N]: the output transform of every LocalAggregation operator
 * (local_aggregation_operators.py:40-45) as streaming kernels.  Training forward = stats (batch statistics ->
 * scale/shift/mean/invstd [C] each, running statistics updated with nn.BatchNorm1d's rule; count = B*N;
 * partial: scratch [cl3d_bn_partials(B,C,N), C, 2] doubles) then apply; inference forward = apply with
 * scale/shift from the running statistics; bwd: coef [5,C] receives A, Bc, D (dx = A dz + Bc + D x), d gamma,
 * d beta; the ReLU mask and xhat are recomputed from x. */
int cl3d_bn_partials(int B, int C, int N);
int cl3d_bn_relu_stats(const float *x, int B, int C, int N, double *partial, int n_partials, double count,
                       float eps, float momentum, const float *gamma, const float *beta, float *running_mean,
                       float *running_var, int64_t *num_batches_tracked, float *scale, float *shift, float *mean,
                       float *invstd, cl3d_stream_t stream);
int cl3d_bn_relu_apply(const float *x, const float *scale, const float *shift, int B, int C, int N, float *out,
                       cl3d_stream_t stream);
int cl3d_bn_relu_bwd(const float *g, const float *x, const float *scale, const float *shift, const float *mean,
                     const float *invstd, const float *gamma, int B, int C, int N, double count, double *partial,
                     int n_partials, float *coef, float *dx, cl3d_stream_t stream);

/* The tail of a bottleneck (backbones/resnet.py:58-66): out = ReLU(BN1(x1) + R) with R = 0 (x2 null), x2 (identity
 * shortcut: scale2 / gamma2 null) or BN2(x2) (shortcut convolution); relu = 0 leaves the sum as it is.  The batch
 * statistics of each branch come from cl3d_bn_relu_stats; apply is one streaming pass.  bwd: g = d out, `out` the saved
 * result (its sign is the ReLU mask); one reduction pass yields both BatchNorm backward sums, coef1 / coef2 [5,C]
 * receive A, Bc, D, d gamma, d beta per branch, dx1 / dx2 the input gradients (dx2 = gated g for the identity);
 * partial: scratch [2, cl3d_bn_partials(B,C,N), C, 2] doubles. */
int cl3d_bn_add_relu_apply(const float *x1, const float *scale1, const float *shift1, const float *x2,
                           const float *scale2, const float *shift2, int relu, int B, int C, int N, float *out,
                           cl3d_stream_t stream);
/* training forward of the same in one call (statistics of each BatchNorm, running-statistics update, apply): vec1 / vec2
 * [4,C] receive scale, shift, mean, invstd -- pass vec + 2C / vec + 3C as mean / invstd to cl3d_bn_add_relu_bwd.  One
 * launch per direction when a channel has <= 16384 values (the deep stages), statistics + apply passes otherwise;
 * partial: [cl3d_bn_partials(B,C,N), C, 2] doubles (only touched on the multi-pass route). */
int cl3d_bn_add_relu_train_fwd(const float *x1, const float *gamma1, const float *beta1, float *running_mean1,
                               float *running_var1, int64_t *num_batches_tracked1, float eps1, float momentum1,
                               const float *x2, const float *gamma2, const float *beta2, float *running_mean2,
                               float *running_var2, int64_t *num_batches_tracked2, float eps2, float momentum2, int relu,
                               int B, int C, int N, double *partial, int n_partials, float *vec1, float *vec2,
                               float *out, cl3d_stream_t stream);
int cl3d_bn_add_relu_bwd(const float *g, const float *out, const float *x1, const float *mean1, const float *invstd1,
                         const float *gamma1, const float *x2, const float *mean2, const float *invstd2,
                         const float *gamma2, int relu, int B, int C, int N, double count, double *partial,
                         int n_partials, float *coef1, float *coef2, float *dx1, float *dx2, cl3d_stream_t stream);

/* [B,R,C] -> [B,C,R] float32: the layout change at the fused operators' boundary (channel-major
 * reference tensors <-> point-major rows). */
int cl3d_transpose(const float *src, int B, int R, int C, float *dst, cl3d_stream_t stream);

/* ---- fused local-aggregation operators (no [B,C,M,K] tensor) -----------------------------------
 * These replace, each in one or a few launches, what the reference's Python does after the grouper
 * in models/local_aggregation_operators.py.  Features and outputs are POINT-MAJOR here:
 * ft [B,N,C], out_t [B,M,C] (the Python layer transposes at the operator boundary). */

/* CSR inverse of a neighbour-index tensor: inv_off [B,N+1], inv_slots [B,MK] (the slot ids of each
 * support point's row, in a fixed order that depends on idx only; slots whose index is outside [0,N)
 * are dropped and inv_off[b][N] counts the kept ones).  ws: cl3d_workspace_bytes(CL3D_OP_INVERSE_INDEX,
 * B, N, M, K, 0).  Used by every fused backward pass: an ordered gather instead of the reference's
 * atomicAdd scatter (group_points_gpu.cu:65), so gradients are bit-reproducible run to run. */
int cl3d_build_inverse_index(const int32_t *idx, int B, int N, int MK, int32_t *inv_off,
                             int32_t *inv_slots, void *ws, size_t ws_bytes, cl3d_stream_t stream);

/* op: 0 PosPool xyz, 1 PosPool sin_cos (p0 = dim table [C/6]), 2 AdaptiveWeight 'dp' with one conv
 * (p0 = W [C/S,3], p1 = bias [C/S], pint = S), 3 PseudoGrid (p0 = K_points [P,3], p1 = kernel_weights
 * [P,C], pint = P, pfloat = 1/extent).  reduction: 0 sum, 1 avg (masked as the reference does,
 * local_aggregation_operators.py:92-103).  Replaces PosPool.forward :65-103, AdaptiveWeight.forward
 * :188-214, PseudoGrid.forward :383-419 up to the output transform.
 * out: [B,M,C], or [B,C,M] (the reference's layout) when out_channel_major != 0.
 * slotrec [B,M,K,4] (nullable) receives what the backward pass needs per slot; pairs [B,M,K,8] (nullable; PseudoGrid
 * with C % 4 == 0 only) the slot's non-zero kernel-point influences {h0..h3, p0..p3}, evaluated once here and read by
 * the backward passes. */
int cl3d_fused_reduce_fwd(int op, const float *query_xyz, const float *support_xyz,
                          const int32_t *query_mask, const int32_t *idx, const int32_t *idx_mask,
                          const float *ft, int B, int N, int M, int K, int C, float radius,
                          int normalize_xyz, int reduction, const float *p0, const float *p1, int pint,
                          float pfloat, int constant_influence, float *out, int out_channel_major,
                          float *slotrec, float *pairs, cl3d_stream_t stream);
/* number of partial blocks of the parameter-gradient buffer dparam [n, C, NP] (NP = 4 adaptive:
 * dW[:,0..2], dbias; 16 pseudo grid: d kernel_weights[p]); 0 for operators without parameters. */
int cl3d_fused_param_partials(int op, int B, int N, int C);
/* dparam [n_partials, C, NP] -> the parameters' gradients, summed in double in a fixed order (what the reference's
 * autograd leaves in conv0.weight.grad / conv0.bias.grad of AdaptiveWeight, local_aggregation_operators.py:188-214,
 * and in kernel_weights.grad of PseudoGrid, :383-419):
 *   CL3D_OP_ADAPTIVE   pint = shared_channels: g0 = d W [C/pint, 3], g1 = d bias [C/pint]
 *   CL3D_OP_PSEUDOGRID pint = kernel points:   g1 = d kernel_weights [pint, C] (g0 unused, may be NULL) */
int cl3d_fused_param_reduce(int op, const float *dparam, int n_partials, int C, int pint, float *g0, float *g1,
                            cl3d_stream_t stream);
int cl3d_fused_reduce_bwd(int op, const float *gout_t, const float *ft, const float *slotrec, const float *pairs,
                          const int32_t *idx, const int32_t *inv_off, const int32_t *inv_slots, int B, int N,
                          int M, int K, int C, const float *p0, const float *p1, int pint, float pfloat,
                          int constant_influence, float *dft, int dft_channel_major, float *dparam,
                          int n_partials, cl3d_stream_t stream);

/* MaskedMaxPool's pooling step (pt_utils.py:194-201: gather + F.max_pool2d over K) without the gathered
 * tensor.  ft [B,N,C] point-major; out [B,C,M] channel-major; kstar_t [B,M,C] = arg-max slot (first
 * maximum) per (query, channel), nullable when no gradient is needed; backward through the CSR inverse. */
int cl3d_maxpool_fwd(const int32_t *idx, const float *ft, int B, int N, int M, int K, int C, float *out,
                     unsigned char *kstar_t, cl3d_stream_t stream);
int cl3d_maxpool_bwd(const float *gout_t, const unsigned char *kstar_t, const int32_t *inv_off,
                     const int32_t *inv_slots, int B, int N, int M, int K, int C, float *dft,
                     int dft_channel_major, cl3d_stream_t stream);
/* The same pooling with the arg-max kept as its SUPPORT INDEX, target_cm int32 [B,C,M] (channel-major like out), and the
 * backward as a scatter with one target per (query, channel): gout [B,C,M] and dfeat [B,C,N] channel-major, no CSR inverse,
 * no transposed gradient; sums in double in LDS (a few floats per row: exact, order-free), every element of dfeat written. */
int cl3d_maxpool_fwd_targets(const int32_t *idx, const float *ft, int B, int N, int M, int K, int C, float *out,
                             int32_t *target_cm, cl3d_stream_t stream);
int cl3d_maxpool_bwd_targets(const float *gout, const int32_t *target_cm, int B, int N, int M, int C, float *dfeat,
                             cl3d_stream_t stream);

/* PointWiseMLP 'dp_fi_df', one Conv2d+BatchNorm2d+ReLU layer, max reduction
 * (local_aggregation_operators.py:288-301).  ght [B,N,2*Co]: row i = [W_d f_i | (W_c - W_d) f_i];
 * wr [Co,3] = the conv weight's columns for the relative position.  See csrc/fused_pwmlp.hip.
 * Training forward = stats -> finalize_stats -> apply; inference forward = fwd (running statistics);
 * backward = bwd_rows -> bwd_hits, bn_backward_coeffs -> bwd_support (+ cl3d_build_inverse_index).
 * Per-(query, channel) arrays are point-major [B,M,Co]; partial buffers are
 * [cl3d_pwmlp_partials(B,M,Co), Co, 8] doubles. */
int cl3d_pwmlp_partials(int B, int M, int Co);

/* The dense contraction of the PointWiseMLP (local_aggregation_operators.py:253-257,288-295) on the matrix cores
 * (csrc/mfma_gemm.hip).  Factored per POINT: ght[b][i] = [W_d f_i | (W_c - W_d) f_i] with W [Co, 3+2C] =
 * [W_r | W_c | W_d] the Conv2d weight; features stay channel-major [B,C,N] as the reference hands them over, ght
 * is point-major [B,N,2Co] (the layout change happens while the tiles go through LDS).  precision: 0 = f32 inputs, v_mfma_f32_32x32x2_f32 (bit-for-bit an fmaf chain); 1 = inputs
 * rounded to bf16 (RNE) while staged, v_mfma_f32_32x32x16_bf16, f32 accumulation and f32 results.
 *   fwd:        ght; also leaves wr [Co,3] = W_r (nullable) and wcat [2Co,C] = [W_d ; W_c - W_d] (one small launch);
 *   bwd_data:   d features [B,C,N] from d ght and the wcat the forward call produced;
 *   bwd_weight: d W [Co, 3+2C] from features, d ght and d wr [Co,3] (nullable: zeros); the sum over all B*N points
 *               is cut into slices summed in a fixed order (bit-reproducible).
 * ws: cl3d_workspace_bytes(CL3D_OP_POINT_GEMM, B, N, Co, 0, C) covers all three; the forward and data-gradient
 * products use it to cut a long contraction into K slices when the output has too few tiles to fill the chip (deep
 * stages) and run unsplit -- same result up to summation order -- when ws is NULL; bwd_weight requires it. */
#define CL3D_PRECISION_F32 0
#define CL3D_PRECISION_BF16 1
int cl3d_pwmlp_point_gemm_fwd(const float *features, const float *W, int B, int C, int N, int Co, int precision,
                              float *ght, float *wr, float *wcat, void *ws, size_t ws_bytes, cl3d_stream_t stream);
int cl3d_pwmlp_point_gemm_bwd_data(const float *dght, const float *wcat, int B, int C, int N, int Co, int precision,
                                   float *dfeatures, void *ws, size_t ws_bytes, cl3d_stream_t stream);
int cl3d_pwmlp_point_gemm_bwd_weight(const float *features, const float *dght, const float *dwr, int B, int C,
                                     int N, int Co, int precision, float *dW, void *ws, size_t ws_bytes,
                                     cl3d_stream_t stream);
/* Both gradients of the per-point product from ONE pass over d ght (round 5, csrc/mfma_gemm.hip
 * pwmlp_point_grads_kernel): a workgroup stages a 64-point tile of d ght and of the features once in LDS and forms
 * d features (wcat^T d ght) and its share of d wcat (d ght^T features) from it; the workgroups' partial tiles are added
 * in workgroup order (bit-reproducible) by the reduce that also writes d W [Co,3+2C].  scale / shift (nullable
 * together): the features enter the weight gradient as max(scale[c] x + shift[c], 0) -- the _pro form below.
 * dfeatures or dW may be NULL (that gradient is not formed).  cl3d_pwmlp_point_gemm_bwd_fused() = 1 where the one-kernel
 * form covers the shape (f32, C <= 64, 2 Co <= 128, whole 64-point tiles per cloud); elsewhere the call runs bwd_data and
 * bwd_weight one after the other on `stream`.  ws: cl3d_workspace_bytes(CL3D_OP_POINT_GEMM, ...), required for dW. */
int cl3d_pwmlp_point_gemm_bwd_fused(int B, int C, int N, int Co, int precision);
int cl3d_pwmlp_point_gemm_bwd(const float *features, const float *scale, const float *shift, const float *dght,
                              const float *wcat, const float *dwr, int B, int C, int N, int Co, int precision,
                              float *dfeatures, float *dW, void *ws, size_t ws_bytes, cl3d_stream_t stream);
/* torch.optim.SGD's update (reference function/train_modelnet_dist.py:137-141: the optimizer of every training loop) over
 * flat buffers of n floats in one launch: g = grad + weight_decay p; buf = first_step ? g : momentum buf + (1 - dampening) g;
 * g = nesterov ? g + momentum buf : buf (momentum_buf non-null iff momentum != 0); p -= lr g; zero_grad != 0: grad = 0
 * afterwards (the flat gradient buffer autograd accumulates into).  closerlook3d_amd/optim.py: FlatSGD. */
int cl3d_sgd_step(float *param, float *grad, float *momentum_buf, long long n, float lr, float momentum,
                  float dampening, float weight_decay, int nesterov, int first_step, int zero_grad,
                  cl3d_stream_t stream);
/* f1 for the PosPool / AdaptiveWeight / PseudoGrid bottlenecks (round 5; backbones/resnet.py:32-39,47-66): the
 * BatchNorm + ReLU of conv1 rides in the layout change that feeds the operator (transpose_bn_relu: [B,R,C] -> [B,C,R]
 * with max(row_scale[r] x + row_shift[r], 0), r = channel), and the operator's own BatchNorm works on the point-major
 * rows [P = B*M, C] the operator writes and conv2 (cl3d_conv1x1_rows_*) reads: bn_rows_stats = batch statistics,
 * scale / shift and the running-statistics update (nn.BatchNorm1d's rule); bn_rows_bwd takes the gradient with respect
 * to the ACTIVATED rows, gates it by the ReLU (recomputed from rows) and returns d rows and coef [5,C] = A, Bc, D,
 * d gamma, d beta.  C % 4 == 0, 16-byte aligned rows; partial [cl3d_bn_rows_partials(P, C), C, 2] doubles. */
int cl3d_transpose_bn_relu(const float *src, const float *row_scale, const float *row_shift, int B, int R, int C,
                           float *dst, cl3d_stream_t stream);
int cl3d_bn_rows_partials(long long P, int C);
int cl3d_bn_rows_stats(const float *rows, long long P, int C, double *partial, int n_partials, double count, float eps,
                       float momentum, const float *gamma, const float *beta, float *running_mean, float *running_var,
                       int64_t *num_batches_tracked, float *scale, float *shift, float *mean, float *invstd,
                       cl3d_stream_t stream);
int cl3d_bn_rows_bwd(const float *g, const float *rows, const float *scale, const float *shift, const float *mean,
                     const float *invstd, const float *gamma, long long P, int C, double count, double *partial,
                     int n_partials, float *coef, float *drows, cl3d_stream_t stream);
/* The 1x1 Conv1d layers either side of the operator (backbones/resnet.py:32-39,58-66; bias-free), same kernel:
 * y [B,Co,N] = W [Co,C] x [B,C,N], its input gradient and its weight gradient
 * (ws: cl3d_workspace_bytes(CL3D_OP_CONV1X1, B, N, Co, 0, C); optional for fwd / bwd_data as above). */
int cl3d_conv1x1_fwd(const float *x, const float *W, int B, int C, int N, int Co, int precision, float *y, void *ws,
                     size_t ws_bytes, cl3d_stream_t stream);
/* inference form: y = act(scale[o] * (W x)[o] + shift[o] + residual) with the eval-mode BatchNorm folded to a
 * per-channel affine map, the shortcut add (residual [B,Co,N], nullable) and the ReLU (relu != 0) in the epilogue;
 * scale / shift nullable together (plain convolution). */
int cl3d_conv1x1_bn_act_fwd(const float *x, const float *W, const float *scale, const float *shift,
                            const float *residual, int relu, int B, int C, int N, int Co, int precision, float *y,
                            void *ws, size_t ws_bytes, cl3d_stream_t stream);
int cl3d_conv1x1_bwd_data(const float *dy, const float *W, int B, int C, int N, int Co, int precision, float *dx,
                          void *ws, size_t ws_bytes, cl3d_stream_t stream);
int cl3d_conv1x1_bwd_weight(const float *x, const float *dy, int B, int C, int N, int Co, int precision, float *dW,
                            void *ws, size_t ws_bytes, cl3d_stream_t stream);

/* A bottleneck without its two [B,C,N] round trips (backbones/resnet.py:32-39,47-66).  The
 * BatchNorm + ReLU that precedes a contraction is applied while the contraction's operand tile is staged
 * (act(x) = max(scale[c] * x + shift[c], 0), scale / shift = the producing layer's folded batch statistics, 16-byte
 * aligned for the vector paths), so the activated tensor between the two layers is never written:
 *   point_gemm_fwd_pro / _bwd_weight_pro: the PointWiseMLP's per-point contraction and its weight gradient on
 *       act(x) with x [B,C,N] = conv1's raw output;
 *   conv1x1_rows_fwd: y [B,Co,N] = W act(x_rows) with x_rows [B,N,C] = the operator's point-major rows (conv2 reading
 *       the operator's output directly); _rows_bwd_data: d x_rows [B,N,C], the gradient with respect to the
 *       ACTIVATED input (what cl3d_pwmlp_bwd_rows takes with gout_channel_major = 0); _rows_bwd_weight: d W [Co,C].
 * ws as for the plain entry points (CL3D_OP_POINT_GEMM / CL3D_OP_CONV1X1). */
int cl3d_pwmlp_point_gemm_fwd_pro(const float *x, const float *scale, const float *shift, const float *W, int B, int C,
                                  int N, int Co, int precision, float *ght, float *wr, float *wcat, void *ws,
                                  size_t ws_bytes, cl3d_stream_t stream);
int cl3d_pwmlp_point_gemm_bwd_weight_pro(const float *x, const float *scale, const float *shift, const float *dght,
                                         const float *dwr, int B, int C, int N, int Co, int precision, float *dW,
                                         void *ws, size_t ws_bytes, cl3d_stream_t stream);
int cl3d_conv1x1_rows_fwd(const float *x_rows, const float *scale, const float *shift, const float *W, int B, int C,
                          int N, int Co, int precision, float *y, void *ws, size_t ws_bytes, cl3d_stream_t stream);
int cl3d_conv1x1_rows_bwd_data(const float *dy, const float *W, int B, int C, int N, int Co, int precision,
                               float *dx_rows, void *ws, size_t ws_bytes, cl3d_stream_t stream);
int cl3d_conv1x1_rows_bwd_weight(const float *x_rows, const float *scale, const float *shift, const float *dy, int B,
                                 int C, int N, int Co, int precision, float *dW, void *ws, size_t ws_bytes,
                                 cl3d_stream_t stream);
/* Tile and K-slice plans of the products above BY MEASUREMENT (process-wide switch, off by default; returns the previous
 * setting).  On: the first call of a product outside stream capture -- key: extents, operand layouts, precision, prologue /
 * epilogue, scratch size -- times the plans the launch-time model prices within 2.5x of its best on the caller's stream
 * (a product is a pure function of its operands) and keeps the winner for the process; inside a capture an unseen product
 * takes the model's plan.  A plan fixes the order in which K slices are summed: with the switch on two PROCESSES may
 * differ in the last bits of a result, one process never does.  cl3d_gemm_autotune_stats: products measured so far,
 * and how many of them kept a plan other than the model's (either pointer may be NULL). */
int cl3d_gemm_autotune(int enable);
int cl3d_gemm_autotune_stats(long long *measured, long long *changed);
/* weight plumbing of the factored contraction: W [Co,3+2C] = [W_r | W_c | W_d] -> wr [Co,3], wcat [2Co,C] =
 * [W_d ; W_c - W_d];  d W from d wr (nullable) and the per-cloud products dwb [B,C,2Co] = F_b G_b. */
int cl3d_pwmlp_split_weight(const float *W, int Co, int C, float *wr, float *wcat, cl3d_stream_t stream);
int cl3d_pwmlp_merge_weight_grad(const float *dwr, const float *dwb, int B, int Co, int C, float *dW,
                                 cl3d_stream_t stream);
/* the training gather pass: per channel sum y, sum y^2, sum y*rel, sum rel (partial); per (query, channel)
 * the extreme pre-activation ystar_t that wins the max (max_k y if gamma >= 0 else min_k y), its slot
 * kstar_t (first one) and sy_t = sum_k y.  Nothing is kept per slot: the backward passes rebuild a slot's relative
 * position from the coordinates and idx. */
int cl3d_pwmlp_stats(const float *query_xyz, const float *support_xyz, const int32_t *idx,
                     const float *ght, const float *wr, const float *gamma, int B, int N, int M, int K,
                     int Co, float radius, float *ystar_t, unsigned char *kstar_t, float *sy_t, double *partial,
                     int n_partials, cl3d_stream_t stream);
/* fixed-order reduction of the double partials + per-channel BatchNorm2d algebra (batch mean/variance,
 * scale/shift, running-statistics update with nn.BatchNorm2d's rule; sums [Co,6] doubles kept for the
 * backward pass).  bn_backward_coeffs: coefficients of dy = A dz [k = k*] + Bc + D y together with
 * d gamma, d beta and d W_r [Co,3]. */
int cl3d_pwmlp_finalize_stats(const double *partial, int n_partials, int Co, double count, float eps,
                              float momentum, const float *gamma, const float *beta, float *running_mean,
                              float *running_var, int64_t *num_batches_tracked, float *scale, float *shift,
                              float *mean, float *invstd, double *sums, cl3d_stream_t stream);
/* out [B,Co,M] = ReLU(scale * ystar + shift)  (== max_k ReLU(BN(y)): the affine map is monotone) */
int cl3d_pwmlp_apply(const float *ystar_t, const float *scale, const float *shift, int B, int M, int Co,
                     float *out, cl3d_stream_t stream);
int cl3d_pwmlp_fwd(const float *query_xyz, const float *support_xyz, const int32_t *idx,
                   const float *ght, const float *wr, const float *scale, const float *shift, int B,
                   int N, int M, int K, int Co, float radius, float *out, int out_channel_major,
                   unsigned char *kstar_t, float *slotrec, cl3d_stream_t stream);
/* dz_cm [B,Co,M] = gout gated by the ReLU at the arg-max and ts_cm [B,Co,M] = idx[j, kstar] (the support point
 * the arg-max slot refers to), both channel-major for bwd_hits; dz_t [B,M,Co] = dz again, point-major, and
 * qtab [B,M,4] (nullable: only cl3d_pwmlp_bwd_support reads it) = {query coordinates, as_float(idx[j, 0])} (one
 * 16-byte record per query: a slot of the slot-walking support pass costs one L2 request instead of three);
 * partial: sum dz, sum dz*xhat, sum dz*rel(k*). */
int cl3d_pwmlp_bwd_rows(const float *gout, int gout_channel_major, const float *ystar_t,
                        const unsigned char *kstar_t, const int32_t *idx, const float *query_xyz,
                        const float *support_xyz, float radius, const float *scale, const float *shift,
                        const float *mean, const float *invstd, int B, int N, int M, int K, int Co, float *dz_cm,
                        int32_t *ts_cm, float *dz_t, float *qtab, double *partial, int n_partials,
                        cl3d_stream_t stream);
/* the arg-max term of d G: hit_cm [B,Co,N] = sum of dz over the (query, channel) pairs with ts = point */
int cl3d_pwmlp_bwd_hits(const float *dz_cm, const int32_t *ts_cm, int B, int N, int M, int Co, float *hit_cm,
                        cl3d_stream_t stream);
int cl3d_pwmlp_bn_backward_coeffs(const double *partial, int n_partials, int Co, double count,
                                  const float *gamma, const float *mean, const float *invstd,
                                  const double *sums, float *cA, float *cB, float *cD, float *dgamma,
                                  float *dbeta, float *dwr, cl3d_stream_t stream);
/* cl3d_pwmlp_bn_backward_coeffs and cl3d_pwmlp_bwd_hits in ONE launch (both only need what cl3d_pwmlp_bwd_rows left;
 * same arguments, same results bit for bit): what the training backward pass calls.  B >= 1. */
int cl3d_pwmlp_bwd_hits_coeffs(const double *partial, int n_partials, double count, const float *gamma,
                               const float *mean, const float *invstd, const double *sums, float *cA, float *cB,
                               float *cD, float *dgamma, float *dbeta, float *dwr, const float *dz_cm,
                               const int32_t *ts_cm, int B, int N, int M, int Co, float *hit_cm,
                               cl3d_stream_t stream);
/* d ght [B,N,2Co] through the CSR inverse of idx (cl3d_build_inverse_index); dz_t, qtab: left by cl3d_pwmlp_bwd_rows of
 * the same step.  Per entry of a support point's slot list one lane looks up the slot's query record in qtab (its
 * coordinates -> the relative position; its centre idx[j, 0], reference local_aggregation_operators.py:290 -> whose H row
 * the slot adds); the row gathers are the only per-(entry, channel) work. */
int cl3d_pwmlp_bwd_support(const float *ght, const float *wr, const float *cA, const float *cB,
                           const float *cD, const float *hit_cm, const float *dz_t, const float *sy_t,
                           const float *qtab, const float *support_xyz, float radius,
                           const int32_t *inv_off, const int32_t *inv_slots, int B, int N, int M, int K, int Co,
                           float *dght, cl3d_stream_t stream);

/* ---- one call per PASS (round 4; csrc/pass.hip).  A PointWiseMLP LocalAggregation in training mode -- ball query, CSR
 * inverse, per-point product, statistics pass, BatchNorm, activation; backward: rows pass, BatchNorm backward, arg-max
 * scatter, support-major pass, both gradient products -- enqueued by ONE call per direction, the geometry work and the
 * weight gradient forked onto streams the library owns and joined by stream-side event waits (the host never waits).
 * What the reference's eager training loop sees is then what it sees of its own extension: one `_ext`-sized call per
 * autograd node (pt_utils.py:16-61).  Every buffer is the caller's; shapes as in the per-kernel entry points above:
 *   idx, idx_mask [B,M,K] (idx_ready != 0: already computed, only read); inv_off [B,N+1] / inv_slots [B,M*K] (NULL: no
 *   backward will follow; csr_ready != 0: already built); bq_ws / csr_ws / gemm_ws*: cl3d_workspace_bytes of
 *   CL3D_OP_BALL_QUERY / _INVERSE_INDEX / _POINT_GEMM; vec [4,Co] = scale, shift, mean, invstd; coef [5,Co] = A, Bc, D,
 *   d gamma, d beta; n_partials = cl3d_pwmlp_partials(B, M, Co) for both partial buffers ([n_partials, Co, 8] doubles).
 * Every fork is joined (stream-side) before the call that made it returns: buffers may be released in stream order. */
typedef struct cl3d_pwmlp_pass {
  int B, N, M, K, C, Co, precision, idx_ready, csr_ready, n_partials;
  float radius, eps, momentum;
  int reserved;  /* (what would be padding: the block has none.  Ignored by the library -- a caller need not zero the
                    block for its passes to be recognised as repeats and replayed from a launch graph) */
  const float *query_xyz, *support_xyz;
  const int32_t *query_mask, *support_mask;
  int32_t *idx, *idx_mask, *inv_off, *inv_slots;
  void *bq_ws, *csr_ws, *gemm_ws, *gemm_ws_d, *gemm_ws_w;
  size_t bq_ws_bytes, csr_ws_bytes, gemm_ws_bytes, gemm_ws_bytes_b;
  const float *features, *W, *gamma, *beta;
  float *running_mean, *running_var;
  int64_t *num_batches_tracked;
  float *ght, *wr, *wcat, *ystar, *sy, *vec, *out;
  unsigned char *kstar;
  double *partial, *sums, *partial_b;
  const float *gout;
  float *dz_cm, *dz_t, *qtab, *hit, *coef, *dwr, *dght, *dfeat, *dW;
  int32_t *ts_cm;
} cl3d_pwmlp_pass;
int cl3d_pwmlp_train_forward(const cl3d_pwmlp_pass *p, cl3d_stream_t stream);
int cl3d_pwmlp_train_backward(const cl3d_pwmlp_pass *p, cl3d_stream_t stream);
/* Launch graphs of the two pass calls (csrc/pass.hip): outside a stream capture, a pass called twice in a row with a
 * bit-identical argument block (an eager training loop in its steady state) is captured into a HIP graph on a stream
 * of the library's and replayed by every later call with that block -- one hipGraphLaunch on the caller's stream instead
 * of a dozen launches; a block seen once is enqueued directly.  _graphs(0) turns this off (returns the previous
 * setting; default on); _graph_stats reports how many passes were captured / replayed since load (nullable). */
int cl3d_pwmlp_pass_graphs(int enable);
int cl3d_pwmlp_pass_graph_stats(long long *captures, long long *replays);

/* ---- the same for the three gather-and-reduce operators (round 6; csrc/pass.hip): a PosPool / AdaptiveWeight / PseudoGrid
 * LocalAggregation up to -- or, optionally, including -- its BatchNorm + ReLU output transform --
 * forward: ball query, layout change of the features (forked), the fused reduction, the CSR inverse (forked behind the
 * query, joined at the end); backward: layout change of the upstream gradient, the support-major pass, the parameters'
 * gradients -- enqueued by ONE call per direction, with the same launch-graph cache as the PointWiseMLP passes
 * (cl3d_pwmlp_pass_graphs / _graph_stats govern and count both).  What it replaces in the reference: MaskedQueryAndGroup
 * + the operator's tensor algebra, local_aggregation_operators.py:65-103,188-214,383-419, one `_ext` call per autograd
 * node in its eager loop (pt_utils.py:16-61).  op / p0 / p1 / pint / pfloat / constant / normalize / reduction as in
 * cl3d_fused_reduce_fwd; features [B,C,N] and out [B,C,M] channel-major (the reference's layout); ft [B,N,C], gout_t
 * [B,M,C] point-major scratch; slotrec [B,M,K,4] / pairs [B,M,K,8] (NULL without a backward / for the other operators);
 * dparam [nparts, C, 4 | 16] with nparts = cl3d_fused_param_partials; g0 / g1: the parameters' gradients
 * (cl3d_fused_param_reduce), NULL for PosPool.  Every buffer is the caller's; every fork is joined before return.
 * Optionally (gamma != NULL) the BatchNorm1d + ReLU output transform rides in the same two calls: see the block's tail. */
typedef struct cl3d_reduce_pass {
  int B, N, M, K, C, op, normalize, reduction, pint, constant, idx_ready, csr_ready, nparts;
  float radius, pfloat;
  int reserved;  /* (what would be padding; ignored, as in cl3d_pwmlp_pass) */
  const float *query_xyz, *support_xyz;
  const int32_t *query_mask, *support_mask;
  const float *features, *p0, *p1;
  int32_t *idx, *idx_mask, *inv_off, *inv_slots;
  void *bq_ws, *csr_ws;
  size_t bq_ws_bytes, csr_ws_bytes;
  float *ft, *out, *slotrec, *pairs;
  const float *gout;
  float *gout_t, *dfeat, *dparam, *g0, *g1;
  /* the operator's output transform when it is the plain BatchNorm1d + ReLU (`out_transform`,
   * local_aggregation_operators.py:43-45,107-110), in the same two calls: gamma != NULL turns it on.  Forward: `out`
   * [B,C,M] is the operator's raw result, `act` [B,C,M] what the caller sees = ReLU(BatchNorm(out)) with batch statistics,
   * running statistics and num_batches_tracked (nullable) updated as nn.BatchNorm1d does; vec [4,C] = scale, shift, mean,
   * invstd; bn_partial [2, bn_parts, C, 2] doubles with bn_parts = cl3d_bn_partials(B, C, M).  Backward: `gout` is the
   * gradient with respect to `act`, graw [B,C,M] scratch, coef [5,C] receives (.., .., .., d gamma, d beta). */
  const float *gamma, *beta;
  float *running_mean, *running_var;
  int64_t *num_batches_tracked;
  float *act, *vec, *graw, *coef;
  double *bn_partial;
  float eps, momentum;
  int bn_parts, reserved2;
} cl3d_reduce_pass;
int cl3d_reduce_train_forward(const cl3d_reduce_pass *p, cl3d_stream_t stream);
int cl3d_reduce_train_backward(const cl3d_reduce_pass *p, cl3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CL3D_H_ */
