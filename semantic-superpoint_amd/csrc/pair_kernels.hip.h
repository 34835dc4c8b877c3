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

// warp_points (utils/utils.py:315-343) of ONE integer point with a PIXEL-space homography P (row-major 9 floats) in the
// reference's own fp32 operation order: torch's CPU `homographies @ points^T` accumulates k = 0, 1, 2 as
// fma(p2, 1, fma(p1, y, p0 * x)) (oneMKL sgemm; verified bit for bit against the oracle for 12 <= N <= 76800 points,
// tests/test_boundary_cpu.py), then the correctly rounded division.  With P computed by the caller exactly like the
// reference (homography_scaling_torch / scale_homography_torch on the host) the warped coordinates - and therefore every
// rounded index derived from them - are bit-identical to the reference's.
__device__ __forceinline__ void warp_point_exact(const float* __restrict__ P, float x, float y, float& wx, float& wy) {
  const float X = __fadd_rn(__fmaf_rn(P[1], y, __fmul_rn(P[0], x)), P[2]);
  const float Y = __fadd_rn(__fmaf_rn(P[4], y, __fmul_rn(P[3], x)), P[5]);
  const float Z = __fadd_rn(__fmaf_rn(P[7], y, __fmul_rn(P[6], x)), P[8]);
  wx = __fdiv_rn(X, Z);
  wy = __fdiv_rn(Y, Z);
}
// pixel-space homography T^-1 H T computed analytically on the device (no host matrix given): same value up to fp32
// rounding of a different operation order, so warped coordinates within ~1e-4 of a .5 boundary may round the other way
__device__ __forceinline__ void pixel_homography_analytic(const float* __restrict__ h, int H, int W, float* __restrict__ P) {
  const float a = 2.f / (float)W, b = 2.f / (float)H;
  float M[9];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    M[r * 3 + 0] = h[r * 3 + 0] * a;
    M[r * 3 + 1] = h[r * 3 + 1] * b;
    M[r * 3 + 2] = -h[r * 3 + 0] - h[r * 3 + 1] + h[r * 3 + 2];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    P[0 * 3 + c] = (M[0 * 3 + c] + M[2 * 3 + c]) / a;
    P[1 * 3 + c] = (M[1 * 3 + c] + M[2 * 3 + c]) / b;
    P[2 * 3 + c] = M[2 * 3 + c];
  }
}


// labels: [B,1,H,W] keypoint map (non-zero = keypoint at integer (x,y)); hn: [B,3,3] normalised homography
// (image -> warped).  out must be zero-filled.  Pixel homography = T^-1 H T, T = [[2/W,0,-1],[0,2/H,-1],[0,0,1]].
__global__ void warp_labels_kernel(const float* __restrict__ labels, const float* __restrict__ hn, const float* __restrict__ hpx,
                                   float* __restrict__ out, int B, int H, int W) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H * W) return;
  if (labels[idx] == 0.f) return;
  const int x = (int)(idx % W), y = (int)((idx / W) % H), n = (int)(idx / ((long)W * H));
  float P[9];
  if (hpx != nullptr) {
#pragma unroll
    for (int k = 0; k < 9; ++k) P[k] = hpx[n * 9 + k];
  } else {
    pixel_homography_analytic(hn + n * 9, H, W, P);
  }
  float wx, wy;
  warp_point_exact(P, (float)x, (float)y, wx, wy);
  if (wx >= 0.f && wx <= (float)(W - 1) && wy >= 0.f && wy <= (float)(H - 1)) {
    const int qx = (int)rintf(wx), qy = (int)rintf(wy);  // torch.round: half to even
    out[((size_t)n * H + qy) * W + qx] = 1.f;
  }
}

// warpLabels(pnts, H, W, homography, bilinear=True) (datasets/data_tools.py:37-63) on a keypoint MAP:
//   out_lab [B,1,H,W]  1 at the rounded warped position of every in-range point
//   out_res [B,2,H,W]  (x, y) residual warped - round(warped), written at the rounded position
//   out_bi  [B,1,H,W]  get_labels_bi (:26-34): the 4 neighbours of the TRUNCATED warped point (all points, the
//                      neighbours are range-filtered individually) receive their bilinear weight
// All three must be zero-filled; scatters are last-write-wins like torch's index_put.
__global__ void warp_labels_full_kernel(const float* __restrict__ labels, const float* __restrict__ hn,
                                        const float* __restrict__ hpx, float* __restrict__ out_lab,
                                        float* __restrict__ out_res, float* __restrict__ out_bi, int B, int H, int W) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * H * W) return;
  if (labels[idx] == 0.f) return;
  const int x = (int)(idx % W), y = (int)((idx / W) % H), n = (int)(idx / ((long)W * H));
  float P[9];
  if (hpx != nullptr) {
#pragma unroll
    for (int k = 0; k < 9; ++k) P[k] = hpx[n * 9 + k];
  } else {
    pixel_homography_analytic(hn + n * 9, H, W, P);
  }
  float wx, wy;
  warp_point_exact(P, (float)x, (float)y, wx, wy);
  const size_t img = (size_t)n * H * W;
  auto inside = [&](float px, float py) { return px >= 0.f && px <= (float)(W - 1) && py >= 0.f && py <= (float)(H - 1); };
  if (out_bi != nullptr && fabsf(wx) < 1e9f && fabsf(wy) < 1e9f) {
    const float xi = truncf(wx), yi = truncf(wy);  // pnts.long(): truncation toward zero
    const float rx = wx - xi, ry = wy - yi;
    if (inside(xi, yi)) out_bi[img + (size_t)yi * W + (size_t)xi] = (1.f - rx) * (1.f - ry);
    if (inside(xi, yi + 1.f)) out_bi[img + (size_t)(yi + 1.f) * W + (size_t)xi] = (1.f - rx) * ry;
    if (inside(xi + 1.f, yi)) out_bi[img + (size_t)yi * W + (size_t)(xi + 1.f)] = rx * (1.f - ry);
    if (inside(xi + 1.f, yi + 1.f)) out_bi[img + (size_t)(yi + 1.f) * W + (size_t)(xi + 1.f)] = rx * ry;
  }
  if (inside(wx, wy)) {
    const float qx = rintf(wx), qy = rintf(wy);  // torch.round: half to even
    const size_t o = (size_t)qy * W + (size_t)qx;
    if (out_lab != nullptr) out_lab[img + o] = 1.f;
    if (out_res != nullptr) {
      out_res[2 * img + o] = wx - qx;
      out_res[2 * img + (size_t)H * W + o] = wy - qy;
    }
  }
}

// datasets/Coco_sem.py:447-448: warped class-id map (float, bilinear) -> int64, pixels outside the valid mask -> n_classes
__global__ void sem_finalize_kernel(const float* __restrict__ sem_w, const float* __restrict__ valid, int64_t* __restrict__ out,
                                    long n, int n_classes) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = valid[i] == 0.f ? (int64_t)n_classes : (int64_t)sem_w[i];
}

// datasets/Coco.py:378,400 `gaussian_blur` = ImgAugTransform (utils/photometric.py:59-78) with GaussianBlur(sigma 0.2): the map goes
// float -> uint8 ((x * 255).astype(np.uint8): truncation) -> blur -> float / 255.  The 5-tap kernel of sigma 0.2 has off-centre
// weights exp(-12.5) = 3.7e-6, which vanish in the 8-bit fixed-point path of the blur: what is left is the quantisation.
__global__ void label_quantize_u8_kernel(const float* __restrict__ in, float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float v = in[i] * 255.f;                                    // float32 product, like numpy
    const float q = v >= 255.f ? 255.f : v > 0.f ? floorf(v) : 0.f;    // astype(uint8) of a value in [0, 255]
    out[i] = q / 255.f;                                                 // float32 division, like numpy
  }
}

// ---- homography sampler (utils/homographies.py:12-141 sample_homography_np + the inversion of datasets/Coco.py:342-350)
// with a counter-based device RNG: distribution-level equivalent of the numpy / scipy streams (parity unpinned, like
// the host generator in synth.py).  One thread per homography.
struct HomographyParams {
  int perspective, scaling, rotation, translation, allow_artifacts;
  int n_scales, n_angles;
  float scaling_amplitude, perspective_amplitude_x, perspective_amplitude_y, patch_ratio, max_angle, translation_overflow;
};

__device__ __forceinline__ uint64_t hs_mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
struct HsRng {
  uint64_t key, ctr;
  __device__ double uniform() { return (double)(hs_mix(key ^ (ctr++ * 0xD1342543DE82EF95ull)) >> 11) * (1.0 / 9007199254740992.0); }
  __device__ double normal() {  // Box-Muller
    const double u1 = fmax(uniform(), 1e-300), u2 = uniform();
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
  }
  __device__ double truncnorm(double std) {  // scipy truncnorm(-2, 2, scale=std) by rejection
    for (;;) {
      const double v = normal();
      if (fabs(v) <= 2.0) return v * std;
    }
  }
};

__global__ void sample_homographies_kernel(uint64_t seed, HomographyParams p, float* __restrict__ out_h,
                                           float* __restrict__ out_inv, int B) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= B) return;
  HsRng rng{hs_mix(seed ^ ((uint64_t)n << 32)), 0};
  const double margin = (1.0 - p.patch_ratio) / 2.0;
  double p2[4][2] = {{margin, margin}, {margin, margin + p.patch_ratio}, {margin + p.patch_ratio, margin + p.patch_ratio},
                     {margin + p.patch_ratio, margin}};
  if (p.perspective) {
    double ax = p.perspective_amplitude_x, ay = p.perspective_amplitude_y;
    if (!p.allow_artifacts) { ax = fmin(ax, margin); ay = fmin(ay, margin); }
    const double pd = rng.truncnorm(ay / 2), hl = rng.truncnorm(ax / 2), hr = rng.truncnorm(ax / 2);
    p2[0][0] += hl; p2[0][1] += pd; p2[1][0] += hl; p2[1][1] -= pd;
    p2[2][0] += hr; p2[2][1] += pd; p2[3][0] += hr; p2[3][1] -= pd;
  }
  auto centre = [&](double& cx, double& cy) {
    cx = 0.25 * (p2[0][0] + p2[1][0] + p2[2][0] + p2[3][0]);
    cy = 0.25 * (p2[0][1] + p2[1][1] + p2[2][1] + p2[3][1]);
  };
  auto in_unit = [&](double q[4][2]) {
    bool ok = true;
    for (int i = 0; i < 4; ++i) ok = ok && q[i][0] >= 0.0 && q[i][0] < 1.0 && q[i][1] >= 0.0 && q[i][1] < 1.0;
    return ok;
  };
  if (p.scaling) {
    double cx, cy;
    centre(cx, cy);
    double scales[17];
    const int ns = min(p.n_scales, 16);
    // utils/homographies.py:83-95: candidates [1, s_1 .. s_n]; with allow_artifacts the index is drawn from arange(n_scales),
    // i.e. over [1, s_1 .. s_{n-1}] (the reference's comment says "all but scale = 1"; its code says this)
    scales[0] = 1.0;
    for (int i = 1; i <= ns; ++i) scales[i] = 1.0 + rng.truncnorm(p.scaling_amplitude / 2);
    int valid[17], nv = 0;
    for (int i = 0; i <= ns; ++i) {
      double q[4][2];
      for (int k = 0; k < 4; ++k) { q[k][0] = (p2[k][0] - cx) * scales[i] + cx; q[k][1] = (p2[k][1] - cy) * scales[i] + cy; }
      if (p.allow_artifacts ? i < ns : in_unit(q)) valid[nv++] = i;
    }
    const double s = nv > 0 ? scales[valid[min((int)(rng.uniform() * nv), nv - 1)]] : 1.0;
    for (int k = 0; k < 4; ++k) { p2[k][0] = (p2[k][0] - cx) * s + cx; p2[k][1] = (p2[k][1] - cy) * s + cy; }
  }
  if (p.translation) {
    double tminx = 1e30, tminy = 1e30, tmaxx = 1e30, tmaxy = 1e30;
    for (int k = 0; k < 4; ++k) {
      tminx = fmin(tminx, p2[k][0]); tminy = fmin(tminy, p2[k][1]);
      tmaxx = fmin(tmaxx, 1.0 - p2[k][0]); tmaxy = fmin(tmaxy, 1.0 - p2[k][1]);
    }
    if (p.allow_artifacts) { tminx += p.translation_overflow; tminy += p.translation_overflow; tmaxx += p.translation_overflow; tmaxy += p.translation_overflow; }
    const double tx = -tminx + rng.uniform() * (tmaxx + tminx), ty = -tminy + rng.uniform() * (tmaxy + tminy);
    for (int k = 0; k < 4; ++k) { p2[k][0] += tx; p2[k][1] += ty; }
  }
  if (p.rotation) {
    double cx, cy;
    centre(cx, cy);
    const int na = min(p.n_angles, 63);
    int valid[64], nv = 0;
    for (int i = 0; i <= na; ++i) {
      const double ang = i < na ? (na > 1 ? -p.max_angle + 2.0 * p.max_angle * i / (na - 1) : 0.0) : 0.0;
      const double c = cos(ang), s = sin(ang);
      double q[4][2];
      for (int k = 0; k < 4; ++k) {  // (pts - centre) @ [[c, -s], [s, c]] + centre
        const double dx = p2[k][0] - cx, dy = p2[k][1] - cy;
        q[k][0] = dx * c + dy * s + cx;
        q[k][1] = -dx * s + dy * c + cy;
      }
      if (p.allow_artifacts ? i < na : in_unit(q)) valid[nv++] = i;  // :118-119: arange(n_angles), the appended 0 is never drawn
    }
    if (nv > 0) {
      const int i = valid[min((int)(rng.uniform() * nv), nv - 1)];
      const double ang = i < na ? (na > 1 ? -p.max_angle + 2.0 * p.max_angle * i / (na - 1) : 0.0) : 0.0;
      const double c = cos(ang), s = sin(ang);
      for (int k = 0; k < 4; ++k) {
        const double dx = p2[k][0] - cx, dy = p2[k][1] - cy;
        p2[k][0] = dx * c + dy * s + cx;
        p2[k][1] = -dx * s + dy * c + cy;
      }
    }
  }
  // pts * shape + shift with shape = (2, 2), shift = -1; solve the 8 x 8 system of the 4 correspondences pts1 -> pts2
  const double p1[4][2] = {{-1, -1}, {-1, 1}, {1, 1}, {1, -1}};
  double A[8][9];
  for (int k = 0; k < 4; ++k) {
    const double x = p1[k][0], y = p1[k][1], u = p2[k][0] * 2.0 - 1.0, v = p2[k][1] * 2.0 - 1.0;
    const double r0[9] = {x, y, 1, 0, 0, 0, -u * x, -u * y, u}, r1[9] = {0, 0, 0, x, y, 1, -v * x, -v * y, v};
    for (int c = 0; c < 9; ++c) { A[2 * k][c] = r0[c]; A[2 * k + 1][c] = r1[c]; }
  }
  for (int c = 0; c < 8; ++c) {  // Gaussian elimination with partial pivoting
    int piv = c;
    for (int r = c + 1; r < 8; ++r) if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
    for (int k = 0; k < 9; ++k) { const double t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
    const double d = A[c][c];
    for (int r = 0; r < 8; ++r) {
      if (r == c) continue;
      const double f = A[r][c] / d;
      for (int k = c; k < 9; ++k) A[r][k] -= f * A[c][k];
    }
  }
  double S[9];
  for (int i = 0; i < 8; ++i) S[i] = A[i][8] / A[i][i];
  S[8] = 1.0;
  // datasets/Coco.py:342-350: the pair uses the INVERSE of the sampled matrix as `homographies`, and its inverse again
  // (= the sampled matrix, re-inverted in fp32 by torch there) as `inv_homographies`
  const double det = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
  const double I[9] = {(S[4] * S[8] - S[5] * S[7]) / det, (S[2] * S[7] - S[1] * S[8]) / det, (S[1] * S[5] - S[2] * S[4]) / det,
                       (S[5] * S[6] - S[3] * S[8]) / det, (S[0] * S[8] - S[2] * S[6]) / det, (S[2] * S[3] - S[0] * S[5]) / det,
                       (S[3] * S[7] - S[4] * S[6]) / det, (S[1] * S[6] - S[0] * S[7]) / det, (S[0] * S[4] - S[1] * S[3]) / det};
  for (int i = 0; i < 9; ++i) {
    out_h[n * 9 + i] = (float)I[i];
    out_inv[n * 9 + i] = (float)S[i];
  }
}

}  // namespace sspk
