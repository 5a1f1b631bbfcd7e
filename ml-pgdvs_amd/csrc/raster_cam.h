// Camera convention shared by the point (A9) and mesh (A10) rasterisers: pytorch3d
// cameras_from_opencv_projection as used by pgdvs/utils/pytorch3d_utils.py:5-47, and the
// pixel <-> NDC mapping of pytorch3d's rasterization_utils (PixToNonSquareNdc).
#pragma once
#include "common.h"

namespace pgdvs {

struct RasterCam {
  float w2c[12];  // first 3 rows of inverse(c2w)
  float fxn, fyn, p0x, p0y;
  // pixel <-> NDC (rasterization_utils PixToNonSquareNdc)
  float range_x, range_y;
};

__device__ __forceinline__ float pix_to_ndc(int i, int S1, float range) {
  float offset = range / 2.0f;
  return -offset + (range * (float)i + offset) / (float)S1;
}

__device__ __forceinline__ RasterCam make_raster_cam(const float *__restrict__ cam, int H, int W) {
  RasterCam rc;
#pragma unroll
  for (int i = 0; i < 12; ++i) rc.w2c[i] = cam[PGDVS_CAM_W2C + i];
  const float fx = cam[PGDVS_CAM_K + 0], fy = cam[PGDVS_CAM_K + 5];
  const float cx = cam[PGDVS_CAM_K + 2], cy = cam[PGDVS_CAM_K + 6];
  float s = (float)(W < H ? W : H) / 2.0f;
  rc.fxn = fx / s;
  rc.fyn = fy / s;
  rc.p0x = -(cx - (float)W / 2.0f) / s;
  rc.p0y = -(cy - (float)H / 2.0f) / s;
  rc.range_x = W > H ? 2.0f * (float)W / (float)H : 2.0f;
  rc.range_y = H > W ? 2.0f * (float)H / (float)W : 2.0f;
  return rc;
}

// world point -> (x_ndc, y_ndc, z_view)
__device__ __forceinline__ float3 point_to_ndc(const RasterCam &rc, float x, float y, float z) {
  float v[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float a = rc.w2c[k * 4 + 0] * x;
    a = a + rc.w2c[k * 4 + 1] * y;
    a = a + rc.w2c[k * 4 + 2] * z;
    a = a + rc.w2c[k * 4 + 3];
    v[k] = a;
  }
  float xv = -v[0], yv = -v[1], zv = v[2];
  float3 r;
  r.x = (rc.fxn * xv + rc.p0x * zv) / zv;
  r.y = (rc.fyn * yv + rc.p0y * zv) / zv;
  r.z = zv;
  return r;
}

}  // namespace pgdvs
