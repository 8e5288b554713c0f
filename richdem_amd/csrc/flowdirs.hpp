// flowdirs.hpp -- internal interface of flowdirs.hip
#pragma once
#include "common.hpp"

namespace rdgpu {
enum { MODE_D8 = 0, MODE_FM = 1 };
// d_dirs[i] in {0 = NO_FLOW, 1..8 = D8 neighbour, 255 = NoData}; mode selects the direction rule.
template <class T>
void flowdirs_device(const T *d_z, T nodata, int w, int h, uint8_t *d_dirs, int mode, hipStream_t s);
}  // namespace rdgpu
