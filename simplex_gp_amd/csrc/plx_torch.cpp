// plx_torch.cpp -- the thin PyTorch-ROCm C++ extension in front of libplx.so.
//
// Drop-in for the reference's pybind11 modules (one symbol, `filter`):
//     gpytorch_lattice_kernel/cuda/permutohedral_cuda.cpp:12-22   filter(src, ref, coeffs) -> Tensor, CHECK_* at :3-5
//     gpytorch_lattice_kernel/cpp/lattice.cpp:6-16                the CPU twin
// as loaded by LatticeFilterGeneral.lazy_compile (bilateral_kernel.py:62-74) and called at py:95 / py:111 / py:119:
//     LatticeFilterGeneral.method = _plx_torch.filter
// Host code only (no kernels here): tensors in, the C ABI of include/plx.h underneath, torch's CURRENT HIP stream
// (the reference launches on the legacy default stream and device-synchronises, cu:476 / cu:524 / cu:543), the GIL
// released around the native call (the reference holds it), errors as Python exceptions (the reference exit()s on a
// CUDA error, cu:24-32).  Also exports the staged pair the reference fuses into every call (h:272): LatticeHandle =
// build once, apply many.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPGuard.h>

#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "plx.h"

namespace {

#define PLX_CHECK_CUDA(x) TORCH_CHECK((x).is_cuda(), #x " must be a CUDA (HIP) tensor")
#define PLX_CHECK_F32(x) TORCH_CHECK((x).scalar_type() == at::kFloat, #x " must be float32 (the reference CPU path is fp32, h:277-278)")
#define PLX_CHECK_2D(x) TORCH_CHECK((x).dim() == 2, #x " must be 2-D, got ", (x).dim(), "-D")

std::vector<float> taps_of(const at::Tensor &coeffs)
{
    TORCH_CHECK(coeffs.dim() == 1 && coeffs.numel() % 2 == 1, "coeffs must be a 1-D tensor with an odd number of taps");
    const at::Tensor t = coeffs.detach().to(at::kCPU, at::kFloat).contiguous();
    return std::vector<float>(t.data_ptr<float>(), t.data_ptr<float>() + t.numel());
}

void check_pair(const at::Tensor &src, const at::Tensor &ref)
{
    PLX_CHECK_CUDA(src); PLX_CHECK_CUDA(ref);
    PLX_CHECK_F32(src); PLX_CHECK_F32(ref);
    PLX_CHECK_2D(src); PLX_CHECK_2D(ref);
    TORCH_CHECK(src.size(0) == ref.size(0), "Incompatible shapes ", src.sizes(), ", and ", ref.sizes());   // py:84-85
    TORCH_CHECK(src.get_device() == ref.get_device(), "src and ref must be on the same device");
}

struct LatticeHandle {
    plx_lattice *lat = nullptr;
    int device = 0;
    at::Tensor ref;            // keeps the positions alive while kernels may still read them
    explicit LatticeHandle(int dev) : device(dev)
    {
        TORCH_CHECK(plx_create(dev, &lat) == PLX_OK, "plx_create: ", plx_last_error());
    }
    ~LatticeHandle() { if (lat) plx_destroy(lat); }
    LatticeHandle(const LatticeHandle &) = delete;
    LatticeHandle &operator=(const LatticeHandle &) = delete;

    void build(const at::Tensor &ref_in, const at::Tensor &coeffs)
    {
        PLX_CHECK_CUDA(ref_in); PLX_CHECK_F32(ref_in); PLX_CHECK_2D(ref_in);
        TORCH_CHECK(ref_in.get_device() == device, "lattice lives on device ", device);
        ref = ref_in.contiguous();
        const std::vector<float> taps = taps_of(coeffs);
        const c10::hip::HIPGuard guard(device);
        void *stream = (void *)c10::hip::getCurrentHIPStream(device).stream();
        int rc;
        {
            pybind11::gil_scoped_release nogil;
            rc = plx_build(lat, ref.data_ptr<float>(), ref.size(0), (int)ref.size(1), taps.data(), (int)taps.size(), 0, 1, stream);
        }
        TORCH_CHECK(rc == PLX_OK, "plx_build: ", plx_last_error());
    }

    at::Tensor apply(const at::Tensor &src_in)
    {
        PLX_CHECK_CUDA(src_in); PLX_CHECK_F32(src_in); PLX_CHECK_2D(src_in);
        TORCH_CHECK(src_in.size(0) == plx_num_points(lat), "Incompatible shapes: ", src_in.sizes(), " for a lattice of ",
                    plx_num_points(lat), " points");
        const at::Tensor src = src_in.contiguous();
        at::Tensor out = at::empty_like(src);
        const c10::hip::HIPGuard guard(device);
        void *stream = (void *)c10::hip::getCurrentHIPStream(device).stream();
        int rc;
        {
            pybind11::gil_scoped_release nogil;
            rc = plx_apply(lat, src.data_ptr<float>(), (int)src.size(1), out.data_ptr<float>(), stream);
        }
        TORCH_CHECK(rc == PLX_OK, "plx_apply: ", plx_last_error());
        return out;
    }

    int64_t num_vertices() const { return plx_num_vertices(lat); }
};

// one scratch lattice per device: its buffers are recycled by every filter() call (plx_filter rebuilds the structure
// each time, like the reference, h:272)
std::mutex g_mutex;
std::unordered_map<int, std::unique_ptr<LatticeHandle>> g_scratch;

at::Tensor filter(at::Tensor src, at::Tensor ref, at::Tensor coeffs)
{
    check_pair(src, ref);
    src = src.contiguous();                                 // the reference passes reference.contiguous() only (py:95)
    ref = ref.contiguous();
    const std::vector<float> taps = taps_of(coeffs);
    const int dev = (int)src.get_device();
    at::Tensor out = at::empty_like(src);
    if (src.size(0) == 0) return out;
    const c10::hip::HIPGuard guard(dev);
    void *stream = (void *)c10::hip::getCurrentHIPStream(dev).stream();
    int rc;
    {
        pybind11::gil_scoped_release nogil;
        std::lock_guard<std::mutex> lock(g_mutex);         // one scratch lattice per device: not for concurrent use
        auto &slot = g_scratch[dev];
        if (!slot) slot.reset(new LatticeHandle(dev));
        rc = plx_filter(slot->lat, src.data_ptr<float>(), ref.data_ptr<float>(), src.size(0), (int)ref.size(1), (int)src.size(1),
                        taps.data(), (int)taps.size(), out.data_ptr<float>(), stream);
    }
    TORCH_CHECK(rc == PLX_OK, "plx_filter: ", plx_last_error());
    return out;
}

}  // namespace

PYBIND11_MODULE(_plx_torch, m)
{
    m.doc() = "MI355X permutohedral-lattice filter (libplx) behind the reference's filter(src, ref, coeffs) boundary";
    m.def("filter", &filter, "filter(src[N,vd], ref[N,d], coeffs[R]) -> out[N,vd]  (cuda/permutohedral_cuda.cpp:12-22)",
          pybind11::arg("src"), pybind11::arg("ref"), pybind11::arg("coeffs"));
    m.def("version", []() { return std::string(plx_version()); });
    pybind11::class_<LatticeHandle>(m, "LatticeHandle")
        .def(pybind11::init<int>(), pybind11::arg("device") = 0)
        .def("build", &LatticeHandle::build, pybind11::arg("ref"), pybind11::arg("coeffs"))
        .def("apply", &LatticeHandle::apply, pybind11::arg("src"))
        .def_property_readonly("num_vertices", &LatticeHandle::num_vertices);
}
