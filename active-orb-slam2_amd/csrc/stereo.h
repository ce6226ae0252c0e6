// Frame::ComputeStereoMatches on gfx950 (reference src/Frame.cc:495-669): internal interface
// between extractor.hip (owns the device pyramids) and stereo.hip (kernels).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/aos2.h"

namespace aos2 {

constexpr int kStereoMaxLevels = 16;

// mvImagePyramid of one eye for a batch of images, as the extractor keeps it in HBM.
struct PyrView {
    const uint8_t *img0;   // level 0 = the (device) input images
    size_t img0_stride;    // bytes between images
    int pitch0;
    const uint8_t *pyr;    // levels >= 1: pyr + image * pyr_bytes + off[level]
    size_t pyr_bytes;
    int nlevels;
    int w[kStereoMaxLevels], h[kStereoMaxLevels], pitch[kStereoMaxLevels];
    size_t off[kStereoMaxLevels];
    float scale[kStereoMaxLevels], inv_scale[kStereoMaxLevels];
};

struct StereoArgs {
    PyrView L, R;
    int first_image_l, first_image_r;   // image index of batch element 0 in each pyramid
    const aos2_keypoint_t *kp_l, *kp_r;  // [batch][cap]
    const uint8_t *desc_l, *desc_r;      // [batch][cap][32]
    const int32_t *n_l, *n_r;            // [batch]
    int cap, batch;
    float mb, mbf;
    float *u_right, *depth;              // [batch][cap]
    int32_t *sad;                        // scratch [batch][cap]
    // vRowIndices (:503-522): for every image row the right keypoints whose band [floor(y - r), ceil(y + r)] covers it
    int rows, row_cap;                   // rows = mvImagePyramid[0].rows; row_cap = entries reserved per image
    int32_t *row_off;                    // scratch [batch][rows + 1]
    int32_t *row_idx;                    // scratch [batch][row_cap]
};

// entries one right keypoint can add to the row table: 2 * ceil(2 * max scale factor) + 3
int stereo_row_span(const PyrView &v);

// enqueue match + cull kernels on `stream`; max_n_left = upper bound of n_l[] (grid size)
int launch_stereo(const StereoArgs &a, int max_n_left, hipStream_t stream);

}  // namespace aos2
