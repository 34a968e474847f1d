// Launch wrappers of the HIP kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "dev_types.h"

namespace jxlhip {

// entropy_kernels.hip (lds_bytes == 0 selects the variant that keeps its tables in global memory)
// LF groups: phase A (ANS, one lane per LF group, one wavefront per image chunk) and phase B (one workgroup per LF group)
void LaunchLfAns(const DevImage* imgs, const SectionTask* tasks, int ntasks, int slots, size_t lds_bytes, int scalar_rows, bool lean, hipStream_t s);
void LaunchLfFinish(const DevImage* imgs, const SectionTask* tasks, int ntasks, hipStream_t s);
void LaunchHfBlockList(const DevImage* imgs, int nimg, int max_groups, hipStream_t s);
size_t HfLaneLdsBytes(int ring_words);
void LaunchHfDecode(const DevImage* imgs, const SectionTask* tasks, int nwg, int threads, int lane_stride, int ring_words, size_t lds_bytes,
                    size_t lane_bytes, hipStream_t s);
void LaunchAlphaAns(const DevImage* imgs, const SectionTask* tasks, int nwg, int lane_stride, size_t lds_bytes, int scalar_rows, bool lean, hipStream_t s);
void LaunchAlphaFinish(const DevImage* imgs, int nimg, int max_groups, hipStream_t s);
// Modular (lossless) frames: per-section ANS phase + predictor phase; inverse transforms (kind 0 RCT, 1 / 2 horizontal / vertical
// unsqueeze of planes a (average), b (residual) into c); clamp + interleave
// lanes: sections per workgroup; rb_width > 0: previous-row buffers (and, with wp_lds, the weighted-predictor state) of the generic
// lane path live in LDS
void LaunchModularAns(const DevImage* imgs, int nimg, const SectionTask* tasks, int ntasks, size_t lds_bytes, int max_sections, int max_coded,
                      int lanes, int rb_width, int wp_lds, int scalar_rows, hipStream_t s);
void LaunchModularOp(int kind, int32_t* a, int32_t* b, int32_t* c, int aw, int ah, int rw, int rh, int type, hipStream_t s);
// inverse Palette: out[k][i] = palette[k * nb_colors + index[i]] for the w x h samples of the index channel; indices outside the
// stored palette (implicit / delta colours) raise the image's error flag
void LaunchModularPalette(const int32_t* palette, const int32_t* index, int32_t* const* out, int nout, int nb_colors, int w, int h, uint32_t* status,
                          hipStream_t s);
void LaunchModularOut(const DevImage* imgs, int nimg, size_t max_pixels, hipStream_t s);
// src: w x h pixels of px_bytes each as stored; dst: the same image with EXIF orientation 2..8 applied (sides swapped for 5..8)
void LaunchOrient(const uint8_t* src, uint8_t* dst, int w, int h, int px_bytes, int orientation, hipStream_t s);
// kernels.hip
void LaunchLfPixelStages(const DevImage* imgs, int nimg, size_t max_cells, hipStream_t s);
// sparse entry lists -> dense int32 coefficient planes, for the listed tiles (generic path) or every tile (debug taps)
void LaunchExpandCoefficients(const DevImage* imgs, int nimg, bool all_tiles, int max_tiles, hipStream_t s);
void LaunchGenericReconstruct(const DevImage* imgs, int nimg, const float* basis_all, const float* basis_small,
                              const float* llf_scale, hipStream_t s);
// tile_kernels.hip
void LaunchReconTiles(const DevImage* imgs, int nimg, int max_tiles, const float* basis_all, const float* basis_small,
                      const float* llf_scale, const float* basis_mfma, hipStream_t s);
void LaunchFilterTiles(const DevImage* imgs, int nimg, int max_w, int max_h, int stage_mask, bool any_unfiltered,
                       int any_fused, int any_fused2, hipStream_t s);

// the batch's status words into pinned host memory, by a kernel (kernels.hip)
void LaunchStatusToHost(const uint32_t* src, uint32_t* dst_pinned, int nwords, hipStream_t s);
#ifdef JXLHIP_EXPERIMENTS
// occupies one kind of resource for `ms` milliseconds on stream s (kernels.hip, interference probes)
void LaunchInterference(int kind, int wg_per_cu, int lds_kb, float ms, float* scratch, size_t scratch_bytes, hipStream_t s);
#endif

}  // namespace jxlhip
