// Internal constants shared by the contrastive kernels.
#pragma once
#include "common.h"

namespace ucd {
// Row granularity of the contrast matrix: the anchor segment and the teacher segment are each padded
// to a multiple of this many rows (padding rows carry label 255 and zero features).
constexpr int kPixTile = 128;
constexpr int kPadLabel = 255;
}  // namespace ucd
