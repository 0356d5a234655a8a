// Internal launcher shared by pillar_scatter.hip and pillar_vfe.hip.
#pragma once
#include "common.h"

namespace dbev {
int launch_canvas(const float* voxel_features, const int* cellmap, float* canvas, int C, int B, int ny, int nx,
                  int channels_last, hipStream_t s);
}
