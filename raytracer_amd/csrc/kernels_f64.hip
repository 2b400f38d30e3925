// kernels_f64.hip — the reference's own arithmetic: f64, every operation rounded separately
// (this file is compiled with -ffp-contract=off).
#include "launch.hpp"
#define RT_REAL double
#include "kernels_impl.inc"

namespace rthip {
hipError_t wf_extend_blocks_per_cu_f64(int threads, bool scene_in_lds, size_t dyn_lds, int* blocks) { return wf_extend_blocks_per_cu_impl(threads, scene_in_lds, dyn_lds, blocks); }
hipError_t render_blocks_per_cu_f64(int threads, int lds_mode, size_t dyn_lds, int* blocks) { return blocks_per_cu_impl(threads, lds_mode, dyn_lds, blocks); }
}  // namespace rthip
