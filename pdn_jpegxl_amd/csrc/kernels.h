// Launch wrappers of the HIP kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "dev_types.h"

namespace jxlhip {

// entropy_kernels.hip (lds_bytes == 0 selects the variant that keeps its tables in global memory)
void LaunchLfGroups(const DevImage* imgs, const SectionTask* tasks, int ntasks, size_t lds_bytes, hipStream_t s);
void LaunchHfDecode(const DevImage* imgs, const SectionTask* tasks, int nwg, int lane_stride, size_t lds_bytes,
                    const uint16_t* natural_orders_small, hipStream_t s);
void LaunchAlpha(const DevImage* imgs, const SectionTask* tasks, int nwg, int lane_stride, size_t lds_bytes, hipStream_t s);
// kernels.hip
void LaunchLfPixelStages(const DevImage* imgs, int nimg, size_t max_cells, hipStream_t s);
void LaunchAlphaToU8(const DevImage* imgs, int nimg, size_t max_pixels, hipStream_t s);
void LaunchReconstruct(const DevImage* imgs, int nimg, size_t max_padded_pixels, size_t max_cells, const float* basis_all,
                       const float* basis_small, const float* llf_scale, hipStream_t s);
void LaunchFiltersAndOutput(const DevImage* imgs, int nimg, size_t max_pixels, bool any_gab, int max_epf, hipStream_t s);

}  // namespace jxlhip
