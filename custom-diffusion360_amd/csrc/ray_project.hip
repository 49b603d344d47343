// A4 + A5 (index part): patch rays, target sample points, projection into every reference view and
// the integer bilinear corner indices.  Standalone (materialising) form of what nerf_fused.hip does
// in registers; used by the `Raymarcher` drop-in, by the un-fused gather path and by the bit-exact
// index parity tests.  Replaces Raymarcher.forward + get_patch_rays (sgm/modules/nerfsd_pytorch3d.py:332-394,
// sgm/modules/utils_cameraray.py:61-196), which bounce every tensor GPU->CPU->GPU.
#include "cd360_geom.h"

namespace {

__global__ void patch_rays_kernel(const float* __restrict__ cams, const float* __restrict__ xs, const float* __restrict__ ys,
                                  float* __restrict__ rays, int ncam_total, int r) {
  const int hw = r * r;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)ncam_total * hw) return;
  const int cam = (int)(gid / hw), k = (int)(gid - (long)cam * hw);
  const Cam cm = load_cam(cams + (long)cam * 16);
  float o[3], d[3];
  patch_ray(cm, xs[k % r], ys[k / r], o, d);
  float* out = rays + gid * 6;
  out[0] = o[0]; out[1] = o[1]; out[2] = o[2]; out[3] = d[0]; out[4] = d[1]; out[5] = d[2];
}

// one thread per (b, view i in 1..n, ray k, sample s)
__global__ void project_index_kernel(const float* __restrict__ cams, const float* __restrict__ xs, const float* __restrict__ ys,
                                     const float* __restrict__ t, int t_ray_stride, int b, int n, int r, int S,
                                     float* __restrict__ points, float* __restrict__ grid, int* __restrict__ x0, int* __restrict__ y0,
                                     int* __restrict__ mask) {
  const int hw = r * r;
  const long total = (long)b * n * hw * S;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int s = (int)(gid % S);
  const int k = (int)((gid / S) % hw);
  const int i = (int)((gid / ((long)S * hw)) % n);
  const int bi = (int)(gid / ((long)S * hw * n));
  const Cam c0 = load_cam(cams + ((long)bi * (n + 1)) * 16);
  const Cam ci = load_cam(cams + ((long)bi * (n + 1) + 1 + i) * 16);
  float o[3], d[3], p[3], v[3];
  patch_ray(c0, xs[k % r], ys[k / r], o, d);
  const float ts = t[(long)k * t_ray_stride + s];
#pragma unroll
  for (int j = 0; j < 3; ++j) p[j] = o[j] + ts * d[j];
  world_to_view(ci, p, v);
  const float gx = grid_coord(ci.f[0], ci.c[0], v[0], v[2]);
  const float gy = grid_coord(ci.f[1], ci.c[1], v[1], v[2]);
  if (points && i == 0) {
    float* pp = points + (((long)bi * hw + k) * S + s) * 3;
    pp[0] = p[0]; pp[1] = p[1]; pp[2] = p[2];
  }
  if (grid) { grid[gid * 2] = gx; grid[gid * 2 + 1] = gy; }
  if (x0 || y0 || mask) {
    const Corner cr = bilinear_corner(gx, gy, r);
    if (x0) x0[gid] = cr.x0;
    if (y0) y0[gid] = cr.y0;
    if (mask) mask[gid] = cr.mask;
  }
}

// Inverse-CDF sampling of depths along a ray (f4: Raymarcher.importance_sampling, nerfsd_pytorch3d.py:264-306, whose
// pytorch3d._C.sample_pdf(bins, weights, outputs, eps) this replaces -- pytorch3d is not in the reference tree; the algorithm is its
// published sample_pdf_python, the NeRF hierarchical sampler): with w_i = weights_i + eps, pdf = w / sum w, cdf = (0, cumsum pdf),
//   k = #{cdf_j <= u} (searchsorted right), below = max(k - 1, 0), above = min(k, n_bins),
//   denom = cdf[above] - cdf[below] (1 if < eps),   sample = bins[below] + (u - cdf[below]) / denom * (bins[above] - bins[below]).
// One thread per (row, sample): the row's n_bins weights are read twice by each of its threads (the same lines for the whole row: L1
// broadcasts) and summed in index order -- the order of a sequential cumsum -- so the result does not depend on the launch shape.
__device__ __forceinline__ float sample_pdf_one(const float* __restrict__ bins, const float* __restrict__ w, float u, float eps, float total,
                                                int n_bins) {
  if (u < 0.f) return bins[0];  // k = 0: below = above = 0
  float c_below = 0.f, c_above = 0.f;
  int k = 1;  // cdf_0 = 0 <= u
  float c = 0.f;
  bool open = true;
  for (int i = 0; i < n_bins; ++i) {
    c += (w[i] + eps) / total;  // cdf_{i+1}
    if (open) {
      if (c <= u) { c_below = c; k = i + 2; }
      else { c_above = c; open = false; }
    }
  }
  const int below = k - 1 < 0 ? 0 : k - 1, above = k > n_bins ? n_bins : k;
  if (open) c_above = c_below;  // u at or beyond the last cdf entry: below = above = n_bins
  float denom = c_above - c_below;
  if (denom < eps) denom = 1.f;
  const float t = (u - c_below) / denom;
  const float b0 = bins[below], b1 = bins[above];
  return b0 + t * (b1 - b0);
}

__global__ void sample_pdf_kernel(const float* __restrict__ bins, const float* __restrict__ weights, const float* u, float* samples,
                                  float* __restrict__ dists, float eps, long rows, int n_bins, int n_samples) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= rows * n_samples) return;
  const long row = gid / n_samples;
  const int j = (int)(gid - row * n_samples);
  const float* w = weights + row * n_bins;
  const float* bn = bins + row * (n_bins + 1);
  float total = 0.f;
  for (int i = 0; i < n_bins; ++i) total += w[i] + eps;
  const float uj = u[gid];
  const float sj = sample_pdf_one(bn, w, uj, eps, total, n_bins);
  if (dists) {  // u and samples are distinct buffers here (checked by the entry): the neighbour's u is still there
    const float nxt = j + 1 < n_samples ? sample_pdf_one(bn, w, u[gid + 1], eps, total, n_bins) : bn[n_bins];
    dists[gid] = nxt - sj;
  }
  samples[gid] = sj;
}

}  // namespace

// bins [rows, n_bins + 1], weights [rows, n_bins], u [rows, n_samples] in [0, 1) -> samples [rows, n_samples] (may alias u: the in-place
// form of pytorch3d._C.sample_pdf) and, optionally, dists [rows, n_samples] = the gaps to the next sample, the last one to the far bin edge
// (nerfsd_pytorch3d.py:306; needs samples != u).  All fp32.
extern "C" int cd360_sample_pdf(const void* bins, const void* weights, const void* u, void* samples, void* dists, float eps, int64_t rows,
                                int n_bins, int n_samples, void* stream) {
  if (!bins || !weights || !u || !samples || rows <= 0 || n_bins <= 0 || n_samples <= 0) return CD360_ERR_ARG;
  if (dists && samples == u) return CD360_ERR_ARG;
  if (!(eps > 0.f)) return CD360_ERR_ARG;
  const long total = (long)rows * n_samples;
  hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)bins,
                     (const float*)weights, (const float*)u, (float*)samples, (float*)dists, eps, (long)rows, n_bins, n_samples);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// cams [b, n+1, 16] fp32; xs, ys [r] NDC patch positions; rays out [b, n+1, r*r, 6] fp32
extern "C" int cd360_patch_rays(const void* cams, const void* xs, const void* ys, void* rays, int b, int n, int r, void* stream) {
  if (!cams || !xs || !ys || !rays || b <= 0 || n < 0 || r <= 0) return CD360_ERR_ARG;
  const long total = (long)b * (n + 1) * r * r;
  hipLaunchKernelGGL(patch_rays_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)cams,
                     (const float*)xs, (const float*)ys, (float*)rays, b * (n + 1), r);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}

// t: sample depths [hw, S] (t_ray_stride = S) or [S] shared by all rays (t_ray_stride = 0).
// outputs (any may be NULL): points [b, hw, S, 3], grid [b, n, hw, S, 2] fp32, x0/y0/mask [b, n, hw, S] int32
extern "C" int cd360_ray_project_index(const void* cams, const void* xs, const void* ys, const void* t, int t_ray_stride, int b, int n,
                                       int r, int S, void* points, void* grid, void* x0, void* y0, void* mask, void* stream) {
  if (!cams || !xs || !ys || !t || b <= 0 || n <= 0 || r <= 0 || S <= 0) return CD360_ERR_ARG;
  if (t_ray_stride != 0 && t_ray_stride != S) return CD360_ERR_SHAPE;
  const long total = (long)b * n * r * r * S;
  hipLaunchKernelGGL(project_index_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)cams, (const float*)xs, (const float*)ys, (const float*)t, t_ray_stride, b, n, r, S, (float*)points,
                     (float*)grid, (int*)x0, (int*)y0, (int*)mask);
  CD360_LAUNCH_CHECK();
  return CD360_OK;
}
