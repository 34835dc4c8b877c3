// Pair construction on the device: the dataset-side warps of the reference
//   inv_warp_image_batch   utils/utils.py:347-385   (bilinear / nearest grid_sample, zeros padding, align_corners=True,
//                                                    grid = linspace(-1,1,W) x linspace(-1,1,H), source = H^-1 * p)
//   compute_valid_mask     utils/utils.py:715-742   (nearest warp of ones + cv2.erode with MORPH_ELLIPSE(2r,2r))
//   warpLabels             datasets/data_tools.py:37-63 (integer keypoints -> T^-1 H T -> keep in range -> round -> scatter 1)
// HBM-bound one-thread-per-pixel kernels; they run once per batch outside the timed step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sspk {

// torch.linspace(-1, 1, n)[i] in fp32 (symmetric evaluation around the midpoint, like ATen)
__device__ __forceinline__ float linspace_m1_1(int i, int n) {
  const float step = 2.0f / (float)(n - 1);
  return (i < n / 2) ? (-1.0f + step * (float)i) : (1.0f - step * (float)(n - 1 - i));
}

// mode 0 = bilinear, 1 = nearest.  img/out: [B,1,H,W]; inv_h: [B,3,3] row-major (normalised coordinates).
__global__ void warp_image_kernel(const float* __restrict__ img, const float* __restrict__ inv_h, float* __restrict__ out,
                                  int B, int H, int W, int mode) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H * W) return;
  const int x = (int)(idx % W), y = (int)((idx / W) % H), n = (int)(idx / ((long)W * H));
  const float* h = inv_h + n * 9;
  const float gx = linspace_m1_1(x, W), gy = linspace_m1_1(y, H);
  const float sx = h[0] * gx + h[1] * gy + h[2];
  const float sy = h[3] * gx + h[4] * gy + h[5];
  const float sw = h[6] * gx + h[7] * gy + h[8];
  const float u = sx / sw, v = sy / sw;
  // grid_sample unnormalise (align_corners=True)
  const float ix = ((u + 1.f) / 2.f) * (float)(W - 1), iy = ((v + 1.f) / 2.f) * (float)(H - 1);
  const float* im = img + (size_t)n * H * W;
  float r = 0.f;
  if (mode == 1) {
    const float fx = nearbyintf(ix), fy = nearbyintf(iy);
    if (fx >= 0.f && fx <= (float)(W - 1) && fy >= 0.f && fy <= (float)(H - 1)) r = im[(int)fy * W + (int)fx];
  } else {
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float ax = ix - x0f, ay = iy - y0f;
    // guard against non-finite coordinates (sw ~ 0): everything outside contributes zero
    if (x0f >= -1.f && x0f <= (float)W && y0f >= -1.f && y0f <= (float)H) {
      const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
      auto at = [&](int yy, int xx) { return (xx >= 0 && xx < W && yy >= 0 && yy < H) ? im[yy * W + xx] : 0.f; };
      // ATen order: nw, ne, sw, se
      r = at(y0, x0) * ((1.f - ax) * (1.f - ay)) + at(y0, x1) * (ax * (1.f - ay)) + at(y1, x0) * ((1.f - ax) * ay) +
          at(y1, x1) * (ax * ay);
    }
  }
  out[idx] = r;
}

// cv2.erode(mask, getStructuringElement(MORPH_ELLIPSE, (2r, 2r))), anchor (r, r), pixels outside the image ignored.
// Ellipse rows: for i in [0, 2r): dy = i - r, dx = round(r * sqrt((r^2 - dy^2) / r^2)), columns [r - dx, r + dx].
__global__ void erode_ellipse_kernel(const float* __restrict__ mask, float* __restrict__ out, int B, int H, int W, int r) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H * W) return;
  const int x = (int)(idx % W), y = (int)((idx / W) % H), n = (int)(idx / ((long)W * H));
  const float* m = mask + (size_t)n * H * W;
  float v = m[y * W + x];
  if (r > 0) {
    const int sz = 2 * r;
    for (int i = 0; i < sz; ++i) {
      const int dy = i - r;
      const int yy = y + dy;
      if (yy < 0 || yy >= H) continue;
      const int dx = (int)lrintf((float)r * sqrtf(fmaxf((float)(r * r - dy * dy), 0.f) / (float)(r * r)));
      const int j0 = max(r - dx, 0), j1 = min(r + dx + 1, sz);
      for (int j = j0; j < j1; ++j) {
        const int xx = x + j - r;
        if (xx >= 0 && xx < W) v = fminf(v, m[yy * W + xx]);
      }
    }
  }
  out[idx] = v;
}

// labels: [B,1,H,W] keypoint map (non-zero = keypoint at integer (x,y)); hn: [B,3,3] normalised homography
// (image -> warped).  out must be zero-filled.  Pixel homography = T^-1 H T, T = [[2/W,0,-1],[0,2/H,-1],[0,0,1]].
__global__ void warp_labels_kernel(const float* __restrict__ labels, const float* __restrict__ hn, float* __restrict__ out,
                                   int B, int H, int W) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H * W) return;
  if (labels[idx] == 0.f) return;
  const int x = (int)(idx % W), y = (int)((idx / W) % H), n = (int)(idx / ((long)W * H));
  const float* h = hn + n * 9;
  const float a = 2.f / (float)W, b = 2.f / (float)H;
  // M = H @ T
  float M[9];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    M[r * 3 + 0] = h[r * 3 + 0] * a;
    M[r * 3 + 1] = h[r * 3 + 1] * b;
    M[r * 3 + 2] = -h[r * 3 + 0] - h[r * 3 + 1] + h[r * 3 + 2];
  }
  // P = T^-1 @ M, T^-1 = [[1/a,0,1/a],[0,1/b,1/b],[0,0,1]]
  float P[9];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    P[0 * 3 + c] = (M[0 * 3 + c] + M[2 * 3 + c]) / a;
    P[1 * 3 + c] = (M[1 * 3 + c] + M[2 * 3 + c]) / b;
    P[2 * 3 + c] = M[2 * 3 + c];
  }
  const float fx = (float)x, fy = (float)y;
  const float X = P[0] * fx + P[1] * fy + P[2], Y = P[3] * fx + P[4] * fy + P[5], Z = P[6] * fx + P[7] * fy + P[8];
  const float wx = X / Z, wy = Y / Z;
  if (wx >= 0.f && wx <= (float)(W - 1) && wy >= 0.f && wy <= (float)(H - 1)) {
    const int qx = (int)rintf(wx), qy = (int)rintf(wy);  // torch.round: half to even
    out[((size_t)n * H + qy) * W + qx] = 1.f;
  }
}

}  // namespace sspk
