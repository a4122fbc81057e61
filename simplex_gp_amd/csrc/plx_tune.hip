// plx_tune.hip -- the switches behind plx_tune(): defaults are the shipped configuration.

#include "plx_kernels.h"

namespace plx {


int g_blur_vpt = 4;     // vertices per thread in the vd = 1 blur (2 or 4)
int g_blur_small = 1;   // all blur passes in one workgroup when m <= 16384 (vd = 1)
int g_xcd_remap = 1;    // 1: workgroup b works on tile (b % 8) * ceil(nb/8) + b / 8, so that the 8 XCDs (which
                               // receive workgroups round-robin) each own one contiguous slice of the lattice
int g_splat_direct = 1;  // vd = 1: gather from d_src through caller-row indices instead of a sorted copy:
                                // 0 never, 1 for launch-bound sizes (<= 2e6 corners: saves a launch; at 9e6 corners
                                // the sorted copy wins, 58 vs 60 us), 2 always
int g_blur_narrow = 1;   // vd 2..16 blur: row length compiled in, branch-free (0: blur_axis_kernel)
int g_blur_multi = 1;    // vd > 1 blur: 4 items per thread (0: one item per thread, blur_axis_kernel)
int g_splat_group = 1;   // vd 2..64: lane-group streaming splat (0: segmented-scan kernel)
int g_splat_wide = 1;    // row-parallel splat for rows of 32..128 chunks (vd 125..512)
#ifdef PLX_DIAG
int g_splat_ablate = 0; // libplx_diag.so only: 1 no value gather, 2 no stores, 4 no row-id loads
int g_blur_ablate = 0;  // libplx_diag.so only: 1 no neighbour gathers, 2 no neighbour-id loads either
#endif

Tunable *tunables()
{
    static Tunable t[] = {{"sort_points", &g_sort_points}, {"order_zcurve", &g_order_zcurve}, {"order_compact", &g_order_compact}, {"readback_spin", &g_readback_spin}, {"vertex_order", &g_vertex_order}, {"insert_plane_fast", &g_insert_plane_fast}, {"compact_nbr", &g_compact_nbr}, {"insert_dedupe", &g_insert_dedupe}, {"nbr_symmetric", &g_nbr_symmetric}, {"blur_vpt", &g_blur_vpt}, {"xcd_remap", &g_xcd_remap}, {"blur_small", &g_blur_small}, {"blur_multi", &g_blur_multi}, {"blur_narrow", &g_blur_narrow}, {"splat_group", &g_splat_group},
                          {"splat_direct", &g_splat_direct}, {"splat_wide", &g_splat_wide}, {"block_path", &g_block_path}, {"block_e", &g_block_e}, {"block_dense_combine", &g_block_dense_combine}, {"blur_fuse", &g_blur_fuse}, {"blur_fuse_vec", &g_blur_fuse_vec}, {"scatter_store", &g_scatter_store}, {"unpermute_gather", &g_unpermute_gather}, 
#ifdef PLX_DIAG
                          {"splat_ablate", &g_splat_ablate}, {"blur_ablate", &g_blur_ablate}, {"block_ablate", &g_block_ablate},
#endif
                          {nullptr, nullptr}};
    return t;
}

}  // namespace plx
