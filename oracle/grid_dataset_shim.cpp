// grid_dataset_shim.cpp -- TEST INFRASTRUCTURE.  A C-ABI caller of the REFERENCE's dataset-side grid
// subsampling (ops/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106), compiled
// together with the reference's own grid_subsampling.cpp and cloud.cpp from where they lie (oracle/build_ref.py,
// `build_grid()`; g++ -std=c++11 as the reference's setup.py, no numpy / Python headers: the reference's
// wrapper.cpp is its CPython binding and is not used).  Nothing of the reference is copied or stood in for;
// this file only marshals flat arrays into the std::vector arguments the reference function takes.
#include <cstring>
#include <vector>

#include "grid_subsampling/grid_subsampling.h"

extern "C" int cl3d_ref_dataset_grid_subsampling(const float *points, const float *features, const int *labels, int n,
                                                 int fdim, int ldim, float sampleDl, float *sub_points,
                                                 float *sub_features, int *sub_labels) {
  std::vector<PointXYZ> in(n), out;
  for (int i = 0; i < n; ++i) in[i] = PointXYZ(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
  std::vector<float> f, sf;
  std::vector<int> l, sl;
  if (features && fdim > 0) f.assign(features, features + (size_t)n * fdim);
  if (labels && ldim > 0) l.assign(labels, labels + (size_t)n * ldim);
  grid_subsampling(in, out, f, sf, l, sl, sampleDl, 0);
  const int m = (int)out.size();
  for (int i = 0; i < m; ++i) {
    sub_points[3 * i] = out[i].x;
    sub_points[3 * i + 1] = out[i].y;
    sub_points[3 * i + 2] = out[i].z;
  }
  if (!sf.empty()) std::memcpy(sub_features, sf.data(), sf.size() * sizeof(float));
  if (!sl.empty()) std::memcpy(sub_labels, sl.data(), sl.size() * sizeof(int));
  return m;  // order = the reference's unordered_map iteration order (implementation-defined)
}
