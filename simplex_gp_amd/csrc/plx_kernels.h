// plx_kernels.h -- device helpers and launch-side switches shared by the per-MVM kernel files
// (plx_splat.hip, plx_blur.hip, plx_slice.hip).  Not part of the C ABI.
//
// Reference: cpp/permutohedral.h ("h") splat value accumulation h:478-479,
// blur h:513-572, slice h:497-510.  The reference's CUDA path does the splat
// with one float atomicAdd per (point, corner, channel) and re-hashes every
// neighbour in every blur pass; here
//   gather-in  right-hand side rows into lattice point order (and, for vd > 1,
//              into rows padded to whole 16-byte vectors),
//   splat      segmented scan over simplex corners sorted by vertex (no
//              atomics, bitwise reproducible),
//   blur       d+1 gather-accumulate passes over a precomputed neighbour table,
//   slice      per-point gather through SoA (vertex id, weight) planes, result
//              scattered back to the caller's row order.
// All are HBM/cache-bandwidth bound gather stencils; no MFMA.
//
// Value rows: vd = 1 -> one float per vertex; vd > 1 -> vdp = roundup4(vd)
// floats, i.e. nch = vdp/4 float4 "chunks", and every access of the vector
// kernels is one aligned 16-byte load/store per lane.
#pragma once

#include "plx_internal.h"

#include <type_traits>

namespace plx {

// Diagnostic ablations (kernels with parts of their memory traffic switched off, for A/B profiles) exist only in
// libplx_diag.so (make diag: -DPLX_DIAG); in the shipped library the switches and the branches behind them are
// compiled out.
#ifdef PLX_DIAG
#define PLX_DIAG_VALUE(x) (x)
#else
#define PLX_DIAG_VALUE(x) 0
#endif

// kernel-variant switches (plx_tune); defined in plx_tune.hip and plx_build.hip

// Tile index for workgroup blockIdx.x.  With remap the launch has 8 * ceil(ntiles / 8) workgroups and
// workgroup b takes tile (b % 8) * per + b / 8: workgroups are dealt to the 8 XCDs round-robin
// (MI355X_MICROARCH.md, Workgroup dispatch), so every XCD sweeps one contiguous eighth of the tiles and its
// gathers -- which follow the lattice order -- stay inside one eighth of the gathered array, i.e. inside
// its own 4 MiB L2.  Placement only affects speed, never results.  Returns -1 for the padding workgroups.
__device__ __forceinline__ int tile_index(int ntiles, int remap)
{
    const int b = blockIdx.x;
    if (!remap) return b < ntiles ? b : -1;
    const int per = (ntiles + 7) >> 3;
    const int t = (b & 7) * per + (b >> 3);
    return ((b >> 3) < per && t < ntiles) ? t : -1;
}
static inline int tile_grid(int ntiles, int remap) { return remap ? 8 * ((ntiles + 7) / 8) : ntiles; }

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_scale(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float4 f4_sel(bool c, float4 a, float4 b) { return c ? a : b; }
__device__ __forceinline__ float4 f4_shfl_up(float4 a, int off)
{
    return make_float4(__shfl_up(a.x, off), __shfl_up(a.y, off), __shfl_up(a.z, off), __shfl_up(a.w, off));
}


template <class V> struct VecOps;
template <> struct VecOps<float> {
    static __device__ __forceinline__ float zero() { return 0.f; }
    static __device__ __forceinline__ float add(float a, float b) { return a + b; }
    static __device__ __forceinline__ float scale(float s, float a) { return s * a; }
    static __device__ __forceinline__ float sel(bool c, float a, float b) { return c ? a : b; }
    static __device__ __forceinline__ float shfl_up(float a, int off) { return __shfl_up(a, off); }
};
template <> struct VecOps<float4> {
    static __device__ __forceinline__ float4 zero() { return f4_zero(); }
    static __device__ __forceinline__ float4 add(float4 a, float4 b) { return f4_add(a, b); }
    static __device__ __forceinline__ float4 scale(float s, float4 a) { return f4_scale(s, a); }
    static __device__ __forceinline__ float4 sel(bool c, float4 a, float4 b) { return f4_sel(c, a, b); }
    static __device__ __forceinline__ float4 shfl_up(float4 a, int off) { return f4_shfl_up(a, off); }
};


}  // namespace plx
