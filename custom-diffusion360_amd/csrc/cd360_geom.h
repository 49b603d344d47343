// Camera / ray geometry shared by ray_project.hip and nerf_fused.hip.
//
// In-kernel replacement for the pytorch3d calls on the pose path
//   cameras.unproject_points / get_camera_center      (sgm/modules/utils_cameraray.py:79-88)
//   ray_bundle_to_ray_points                          (sgm/modules/nerfsd_pytorch3d.py:381-387)
//   cam.transform_points_ndc                          (sgm/modules/nerfsd_pytorch3d.py:73-77)
// and of the coordinate part of F.grid_sample(align_corners=True) (nerfsd_pytorch3d.py:79-98).
//
// Every expression is an ordered chain of single fp32 roundings, identical to oracle/pose_path.py
// (world_to_view, patch_rays, project_ndc, sample_grid, bilinear_corners).  The translation units
// that include this header are compiled with -ffp-contract=off; the one fused operation the CPU
// path has (torch's 3-element norm = sqrt(fma(z,z,fma(y,y,x*x)))) is written with explicit fmaf.
// Division and sqrt are IEEE-correct (hipcc default).  Result: the integer bilinear corner
// indices are bit-exact against the oracle.
#pragma once
#include "cd360_common.h"

struct Cam {  // packed row: R (row-major 9) | T 3 | focal 2 | principal point 2
  float R[9], T[3], f[2], c[2];
};

__device__ __forceinline__ Cam load_cam(const float* p) {
  Cam cm;
#pragma unroll
  for (int i = 0; i < 9; ++i) cm.R[i] = p[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) cm.T[i] = p[9 + i];
  cm.f[0] = p[12]; cm.f[1] = p[13]; cm.c[0] = p[14]; cm.c[1] = p[15];
  return cm;
}

// X_view = X_world @ R + T
__device__ __forceinline__ void world_to_view(const Cam& cm, const float p[3], float v[3]) {
#pragma unroll
  for (int j = 0; j < 3; ++j) v[j] = ((p[0] * cm.R[0 + j] + p[1] * cm.R[3 + j]) + p[2] * cm.R[6 + j]) + cm.T[j];
}
// d @ R
__device__ __forceinline__ void rotate_to_view(const Cam& cm, const float d[3], float v[3]) {
#pragma unroll
  for (int j = 0; j < 3; ++j) v[j] = (d[0] * cm.R[0 + j] + d[1] * cm.R[3 + j]) + d[2] * cm.R[6 + j];
}
// C = -T @ R^T
__device__ __forceinline__ void camera_center(const Cam& cm, float o[3]) {
  const float n0 = -cm.T[0], n1 = -cm.T[1], n2 = -cm.T[2];
#pragma unroll
  for (int j = 0; j < 3; ++j) o[j] = (n0 * cm.R[3 * j + 0] + n1 * cm.R[3 * j + 1]) + n2 * cm.R[3 * j + 2];
}
// ray through NDC (x, y): origin + unit direction in world space
__device__ __forceinline__ void patch_ray(const Cam& cm, float x, float y, float o[3], float d[3]) {
  const float xv = ((x - cm.c[0]) * 1.0f) / cm.f[0];
  const float yv = ((y - cm.c[1]) * 1.0f) / cm.f[1];
  const float a0 = xv - cm.T[0], a1 = yv - cm.T[1], a2 = 1.0f - cm.T[2];
  camera_center(cm, o);
#pragma unroll
  for (int j = 0; j < 3; ++j) d[j] = ((a0 * cm.R[3 * j + 0] + a1 * cm.R[3 * j + 1]) + a2 * cm.R[3 * j + 2]) - o[j];
  const float nrm = sqrtf(fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0])));
#pragma unroll
  for (int j = 0; j < 3; ++j) d[j] = d[j] / nrm;
}
// grid_sample coordinate of a view-space point: clip(nan_to_num(-ndc), -1.2, 1.2)
__device__ __forceinline__ float grid_coord(float f, float c, float vxy, float vz) {
  float g = -((f * vxy) / vz + c);
  if (g != g) g = 0.f;
  return fminf(fmaxf(g, -1.2f), 1.2f);
}

struct Corner {
  int x0, y0;     // north-west texel
  float tx, ty;   // fractional weights towards east / south
  int mask;       // bit0 nw, bit1 ne, bit2 sw, bit3 se in-bounds
};
__device__ __forceinline__ Corner bilinear_corner(float gx, float gy, int r) {
  const float sc = (float)(r - 1);
  const float ix = ((gx + 1.0f) * 0.5f) * sc, iy = ((gy + 1.0f) * 0.5f) * sc;
  const float fx = floorf(ix), fy = floorf(iy);
  Corner cr;
  cr.x0 = (int)fx; cr.y0 = (int)fy;
  cr.tx = ix - fx; cr.ty = iy - fy;
  const bool x0ok = cr.x0 >= 0 && cr.x0 < r, x1ok = cr.x0 + 1 >= 0 && cr.x0 + 1 < r;
  const bool y0ok = cr.y0 >= 0 && cr.y0 < r, y1ok = cr.y0 + 1 >= 0 && cr.y0 + 1 < r;
  cr.mask = (int)(x0ok && y0ok) | ((int)(x1ok && y0ok) << 1) | ((int)(x0ok && y1ok) << 2) | ((int)(x1ok && y1ok) << 3);
  return cr;
}
