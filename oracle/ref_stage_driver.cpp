// TEST INFRASTRUCTURE ONLY -- stage dump of the REFERENCE's CPU lattice.
//
// This file is ours; it #includes the reference header where it lies
// (/root/reference/gpytorch_lattice_kernel/cpp/permutohedral.h, passed with -I by
// oracle/build_ref.py) and drives its public members (splat / blur / slice,
// hashTable.getKeys / getValues / size, greedy, rank: h:587-590) to record what
// the reference computes between the stages of filter() (h:259-340).  Used only
// by tests/golden/make_golden.py in the dev container.
#include <chrono>
#include <cassert>
#include <torch/extension.h>
#include "permutohedral.h"

// returns {keys [m,d] int16, values_after_splat [m,vd], values_after_blur [m,vd], out [n,vd],
//          greedy [n,d+1] int16, rank [n,d+1] int8}
std::vector<at::Tensor> stages(at::Tensor src, at::Tensor ref, at::Tensor coeffs)
{
    const int n = src.size(0), vd = src.size(1), d = ref.size(1);
    src = src.contiguous();
    ref = ref.contiguous();
    PermutohedralLattice lattice(d, vd, n, coeffs);
    auto greedy = torch::empty({n, d + 1}, torch::kInt16);
    auto rank = torch::empty({n, d + 1}, torch::kInt8);
    for (int i = 0; i < n; ++i) {
        lattice.splat(ref.data_ptr<float>() + (size_t)i * d, src.data_ptr<float>() + (size_t)i * vd);
        for (int j = 0; j <= d; ++j) {
            greedy.data_ptr<int16_t>()[(size_t)i * (d + 1) + j] = lattice.greedy[j];
            rank.data_ptr<int8_t>()[(size_t)i * (d + 1) + j] = lattice.rank[j];
        }
    }
    const int m = lattice.hashTable.size();
    auto keys = torch::empty({m, d}, torch::kInt16);
    memcpy(keys.data_ptr<int16_t>(), lattice.hashTable.getKeys(), sizeof(short) * (size_t)m * d);
    auto v_splat = torch::empty({m, vd}, torch::kFloat32);
    memcpy(v_splat.data_ptr<float>(), lattice.hashTable.getValues(), sizeof(float) * (size_t)m * vd);
    lattice.blur(coeffs);
    auto v_blur = torch::empty({m, vd}, torch::kFloat32);
    memcpy(v_blur.data_ptr<float>(), lattice.hashTable.getValues(), sizeof(float) * (size_t)m * vd);
    lattice.beginSlice();
    auto out = torch::zeros({n, vd}, torch::kFloat32);
    for (int i = 0; i < n; ++i) lattice.slice(out.data_ptr<float>() + (size_t)i * vd);
    return {keys, v_splat, v_blur, out, greedy, rank};
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) { m.def("stages", &stages, "stage dump of the reference CPU lattice"); }
