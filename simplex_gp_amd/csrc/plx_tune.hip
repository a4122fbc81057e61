// plx_tune.hip -- the switches behind plx_tune(): the process-wide defaults (the shipped configuration) and the
// name table.  Lattices work under a snapshot of the defaults taken when their build starts (Tune in plx_internal.h).

#include "plx_kernels.h"

namespace plx {

Tune g_tune_defaults;
thread_local const Tune *tl_tune = &g_tune_defaults;

const Tunable *tunables()
{
    static const Tunable t[] = {{"sort_points", &Tune::sort_points}, {"order_zcurve", &Tune::order_zcurve}, {"order_compact", &Tune::order_compact},
                                {"readback_spin", &Tune::readback_spin}, {"vertex_order", &Tune::vertex_order}, {"insert_plane_fast", &Tune::insert_plane_fast},
                                {"compact_nbr", &Tune::compact_nbr}, {"insert_dedupe", &Tune::insert_dedupe}, {"nbr_symmetric", &Tune::nbr_symmetric},
                                {"blur_vpt", &Tune::blur_vpt}, {"xcd_remap", &Tune::xcd_remap}, {"blur_small", &Tune::blur_small}, {"blur_multi", &Tune::blur_multi},
                                {"blur_narrow", &Tune::blur_narrow}, {"splat_group", &Tune::splat_group}, {"splat_direct", &Tune::splat_direct},
                                {"splat_wide", &Tune::splat_wide}, {"block_path", &Tune::block_path}, {"block_e", &Tune::block_e},
                                {"block_dense_combine", &Tune::block_dense_combine}, {"blur_fuse", &Tune::blur_fuse}, {"blur_fuse_vec", &Tune::blur_fuse_vec},
                                {"nbr_bitmap", &Tune::nbr_bitmap}, {"nbr_window", &Tune::nbr_window}, {"perm_rows", &Tune::perm_rows}, {"splat_first", &Tune::splat_first}, {"blk_sort", &Tune::blk_sort}, {"reference_growth", &Tune::reference_growth}, {"embed_vrange", &Tune::embed_vrange}, {"order_sample", &Tune::order_sample}, {"insert_xcd", &Tune::insert_xcd}, {"assign_evid", &Tune::assign_evid}, {"nbr_seed", &Tune::nbr_seed}, {"nbr_sliced", &Tune::nbr_sliced}, {"scatter_store", &Tune::scatter_store}, {"unpermute_gather", &Tune::unpermute_gather}, {"contract_v", &Tune::contract_v}, {"blur_active", &Tune::blur_active},
#ifdef PLX_DIAG
                                {"splat_ablate", &Tune::splat_ablate}, {"blur_ablate", &Tune::blur_ablate}, {"block_ablate", &Tune::block_ablate},
#endif
                                {nullptr, nullptr}};
    return t;
}

}  // namespace plx
