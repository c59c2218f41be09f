// Corner selection of the warped-flow step on the host (cv::goodFeaturesToTrack's sequential half): host-only, sanitizer-built.
#pragma once
#include <cstddef>

namespace vq {
// peaks [h][w]: corner strength at the 3x3 local maxima, 0 elsewhere; top: the frame's largest strength.  Returns the number of
// corners written to out_xy (x, y pairs, strongest first).
int select_corners(const float* peaks, int h, int w, float top, int max_corners, float quality, float min_distance, float* out_xy);
// n frames over host threads (frame p -> thread p % workers); top_bits[p] = the bit pattern of frame p's largest strength
void select_corners_batch(const float* peaks, const unsigned* top_bits, int n, int h, int w, int max_corners, float quality, float min_distance,
                          float* corners_xy, int* counts);
}  // namespace vq
