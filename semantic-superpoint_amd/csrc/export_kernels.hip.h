// Homography-adaptation export: the per-image body of
// export_detector_homoAdapt_gpu (export.py:274-318) after the network forward.
//   homoadapt_views_kernel     datasets/Coco.py:258-292  n warped copies of ONE image + their nearest-warped masks
//   flatten_detection_kernel   utils/utils.py:515-560    softmax over 65 channels, drop dustbin, DepthToSpace(8)
//                                                        (utils/d2s.py:8-27), times the view's valid mask (export.py:51)
//   combine_heatmap_kernel     export.py:49-60           bilinear un-warp of heatmap*mask and mask per view, sum over
//                                                        the views, divide (0/0 stays NaN)
//   nms_init / nms_points      models/model_wrap.py:266-293,129-192 threshold, greedy grid NMS, border removal,
//                                                        descending sort; models/model_wrap.py:212-249 5x5 soft-argmax
//                                                        refinement; export.py:303-309 top-k
// All of it is HBM/latency-bound index and gather work: one thread per pixel, no MFMA.
//
// Greedy NMS in parallel form.  nms_fast visits the candidates by descending confidence and keeps one iff no
// already-kept point lies within Chebyshev distance d.  Equivalent fixed point: a candidate is decided as soon as all
// its higher-priority neighbours are decided -- KEPT if none of them was kept, SUPPRESSED otherwise.  Rounds of
// "every undecided candidate looks at its (2d+1)^2 window" reach exactly the sequential result (priority = confidence,
// ties broken by the lower row-major index, i.e. a stable sort of np.where order).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pair_kernels.hip.h"

namespace sspk {

// grid_sample(bilinear, zeros padding, align_corners=True) of one [H,W] plane at normalised (u, v)
__device__ __forceinline__ float bilinear_zero(const float* __restrict__ im, int H, int W, float u, float v) {
  const float ix = ((u + 1.f) / 2.f) * (float)(W - 1), iy = ((v + 1.f) / 2.f) * (float)(H - 1);
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float ax = ix - x0f, ay = iy - y0f;
  if (!(x0f >= -1.f && x0f <= (float)W && y0f >= -1.f && y0f <= (float)H)) return 0.f;
  const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
  auto at = [&](int yy, int xx) { return (xx >= 0 && xx < W && yy >= 0 && yy < H) ? im[yy * W + xx] : 0.f; };
  return at(y0, x0) * ((1.f - ax) * (1.f - ay)) + at(y0, x1) * (ax * (1.f - ay)) + at(y1, x0) * ((1.f - ax) * ay) +
         at(y1, x1) * (ax * ay);
}

// img: [H,W]; inv_h: [n,3,3]; views, masks: [n,H,W].  masks = nearest warp of an all-ones image (erosion is a
// separate erode_ellipse_kernel launch when the radius is non-zero).
__global__ void homoadapt_views_kernel(const float* __restrict__ img, const float* __restrict__ inv_h,
                                       float* __restrict__ views, float* __restrict__ masks, int n, int H, int W) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * H * W) return;
  const int x = (int)(idx % W), y = (int)((idx / W) % H), v = (int)(idx / ((long)W * H));
  const float* h = inv_h + v * 9;
  const float gx = linspace_m1_1(x, W), gy = linspace_m1_1(y, H);
  const float sw = h[6] * gx + h[7] * gy + h[8];
  const float su = (h[0] * gx + h[1] * gy + h[2]) / sw, sv = (h[3] * gx + h[4] * gy + h[5]) / sw;
  views[idx] = bilinear_zero(img, H, W, su, sv);
  const float fx = nearbyintf(((su + 1.f) / 2.f) * (float)(W - 1)), fy = nearbyintf(((sv + 1.f) / 2.f) * (float)(H - 1));
  masks[idx] = (fx >= 0.f && fx <= (float)(W - 1) && fy >= 0.f && fy <= (float)(H - 1)) ? 1.f : 0.f;
}

// One wave per cell: lane c holds channel c (pixel (c/8, c%8) of the cell); the dustbin (channel 64) is read by all.
// y: detector logits, element (n, cell, c) at n*img_stride + cell*cell_stride + c*chan_stride -- the engine's raw
// convPb output (NHWC, bnPb applied here through scale/shift) or a public NCHW `semi` tensor (scale == null).
// heat[n, 8*hc + c/8, 8*wc + c%8] = softmax(semi)[c] * mask (mask may be null).
__global__ __launch_bounds__(256) void flatten_detection_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mask, float* __restrict__ heat,
                                                                int ncells, int Hc, int Wc, long img_stride,
                                                                long cell_stride, long chan_stride) {
  const int cell = blockIdx.x * 4 + (threadIdx.x >> 6), c = threadIdx.x & 63;
  if (cell >= ncells) return;
  const float* p = y + (size_t)(cell / (Hc * Wc)) * img_stride + (size_t)(cell % (Hc * Wc)) * cell_stride;
  float v = p[c * chan_stride], dust = p[64 * chan_stride];
  if (scale) {
    v = v * scale[c] + shift[c];
    dust = dust * scale[64] + shift[64];
  }
  float m = v;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  m = fmaxf(m, dust);
  const float e = expf(v - m);
  float s = e;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  s += expf(dust - m);
  const int wc = cell % Wc, hc = (cell / Wc) % Hc, n = cell / (Wc * Hc);
  const size_t idx = ((size_t)n * Hc * 8 + hc * 8 + (c >> 3)) * (size_t)(Wc * 8) + wc * 8 + (c & 7);
  heat[idx] = (e / s) * (mask ? mask[idx] : 1.f);
}

// heat (already multiplied by mask), mask: [n,H,W]; hm: [n,3,3] un-warp matrices; out: [H,W].
// block = 64 pixels x 4 view groups; the 4 partial sums are combined in a fixed order.
__global__ __launch_bounds__(256) void combine_heatmap_kernel(const float* __restrict__ heat, const float* __restrict__ mask,
                                                              const float* __restrict__ hm, float* __restrict__ out,
                                                              int n, int H, int W) {
  __shared__ float sh[2][4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int pix = blockIdx.x * 64 + lane;
  const bool live = pix < H * W;
  const int x = live ? pix % W : 0, y = live ? pix / W : 0;
  const float gx = linspace_m1_1(x, W), gy = linspace_m1_1(y, H);
  float ah = 0.f, am = 0.f;
  for (int v = grp; v < n; v += 4) {
    const float* h = hm + v * 9;
    const float sw = h[6] * gx + h[7] * gy + h[8];
    const float su = (h[0] * gx + h[1] * gy + h[2]) / sw, sv = (h[3] * gx + h[4] * gy + h[5]) / sw;
    ah += bilinear_zero(heat + (size_t)v * H * W, H, W, su, sv);
    am += bilinear_zero(mask + (size_t)v * H * W, H, W, su, sv);
  }
  sh[0][grp][lane] = ah;
  sh[1][grp][lane] = am;
  __syncthreads();
  if (grp == 0 && live) {
    const float a = ((sh[0][0][lane] + sh[0][1][lane]) + sh[0][2][lane]) + sh[0][3][lane];
    const float b = ((sh[1][0][lane] + sh[1][1][lane]) + sh[1][2][lane]) + sh[1][3][lane];
    out[pix] = a / b;
  }
}

// ------------------------------------------------------------------------------------------------
// points from one aggregated heatmap
// ------------------------------------------------------------------------------------------------
struct PointsWork {       // device scratch of one image (ssp_export_workspace_bytes)
  uint8_t* state;         // [H*W] 0 = empty / suppressed, 1 = undecided, 2 = kept
  int32_t* cand[2];       // [H*W] undecided lists (ping-pong)
  uint64_t* keys;         // [cap2] sort keys of the kept points, cap2 = power of two
  int32_t* counters;      // [1] = number of kept points inside the border band
};

enum { ST_EMPTY = 0, ST_UNDECIDED = 1, ST_KEPT = 2 };

// 5x5 soft-argmax around pixel (x, y) of the zero-padded heatmap: utils/losses.py:64-91 extract_patch_from_points,
// :53-61 norm_patches (sum + 1e-6), :138-142 do_log, torchgeometry contrib.SpatialSoftArgmax2d (softmax with max
// subtraction, 1/(sum + 1e-6), un-normalised coordinates 0..4).  (sx, sy) = expected (column, row) in [0, 4].
__device__ __forceinline__ void soft_argmax5(const float* __restrict__ heat, int H, int W, int x, int y, float& sx,
                                             float& sy) {
  float p[25];
  float sum = 0.f;
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      const int yy = y + a - 2, xx = x + b - 2;
      const float q = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? heat[yy * W + xx] : 0.f;
      p[a * 5 + b] = q;
      sum += q;
    }
  const float d = sum + 1e-6f;
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    float q = p[t] / d;
    if (q < 0.f) q = 1e-6f;
    p[t] = logf(q);
    m = fmaxf(m, p[t]);
  }
  float es = 0.f;
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    p[t] = expf(p[t] - m);
    es += p[t];
  }
  const float inv = 1.f / (es + 1e-6f);
  sx = 0.f; sy = 0.f;
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    sx += ((float)(t % 5) * p[t]) * inv;
    sy += ((float)(t / 5) * p[t]) * inv;
  }
}

// xy: [n][2] float (x, y), truncated to int like `points.astype(int)`; out: [n][2] (sx, sy)
__global__ void soft_argmax_points_kernel(const float* __restrict__ heat, const float* __restrict__ xy, float* __restrict__ out,
                                          int n, int H, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float sx, sy;
  soft_argmax5(heat, H, W, (int)xy[2 * i], (int)xy[2 * i + 1], sx, sy);
  out[2 * i] = sx;
  out[2 * i + 1] = sy;
}

__device__ __forceinline__ uint8_t ld_state(const uint8_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_state(uint8_t* p, uint8_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a kept point inside the border band still suppresses its neighbours but is not exported (model_wrap.py:286-292)
__device__ __forceinline__ void push_kept(const PointsWork& w, int i, int x, int y, float v, int H, int W, int border,
                                          int cap2) {
  if (x >= border && x < W - border && y >= border && y < H - border) {
    const int pos = atomicAdd(w.counters + 1, 1);
    if (pos < cap2) w.keys[pos] = ((uint64_t)__float_as_uint(v) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
  }
}

// state[i] = heat[i] >= thresh (NaN compares false); also clears the sort keys
__global__ __launch_bounds__(256) void nms_init_kernel(const float* __restrict__ heat, float thresh, PointsWork w, int HW,
                                                       int cap2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < cap2) w.keys[i] = 0ull;
  if (i >= HW) return;
  w.state[i] = heat[i] >= thresh ? ST_UNDECIDED : ST_EMPTY;
}

// Tiled rounds: one block per 32x32 tile, one thread per pixel, the tile and a halo of `dist` pixels (values + states)
// in LDS.  A block iterates until its own candidates are decided; halo candidates belong to neighbouring blocks and
// are re-read from HBM every pass.  States only ever move UNDECIDED -> {EMPTY, KEPT} and a candidate is decided only
// once all its higher-priority neighbours are, so stale halo reads merely delay a decision.  A block that makes no
// progress for `max_idle` passes (its neighbours are not resident) gives up; nms_points_kernel finishes the rest.
constexpr int NMS_TILE = 32, NMS_MAX_HALO = 8, NMS_PITCH_MAX = NMS_TILE + 2 * NMS_MAX_HALO;

__global__ __launch_bounds__(1024) void nms_tiles_kernel(const float* __restrict__ heat, PointsWork w, int H, int W, int dist,
                                                         int border, int cap2, int max_idle) {
  __shared__ float sv[NMS_PITCH_MAX * NMS_PITCH_MAX];
  __shared__ uint8_t ss[NMS_PITCH_MAX * NMS_PITCH_MAX];
  __shared__ int pending, progress;
  const int tid = threadIdx.x;
  const int tiles_x = (W + NMS_TILE - 1) / NMS_TILE;
  const int tx0 = (blockIdx.x % tiles_x) * NMS_TILE, ty0 = (blockIdx.x / tiles_x) * NMS_TILE;
  const int P = NMS_TILE + 2 * dist;
  for (int k = tid; k < P * P; k += 1024) {
    const int gy = ty0 - dist + k / P, gx = tx0 - dist + k % P;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    sv[k] = in ? heat[gy * W + gx] : 0.f;
    ss[k] = in ? ld_state(w.state + gy * W + gx) : (uint8_t)ST_EMPTY;
  }
  const int lx = tid & 31, ly = tid >> 5, gx = tx0 + lx, gy = ty0 + ly;
  const int c = (ly + dist) * P + lx + dist, gi = gy * W + gx;
  __syncthreads();
  bool mine = gx < W && gy < H && ss[c] == ST_UNDECIDED;
  const float v = sv[c];
  int idle = 0;
  for (;;) {
    __syncthreads();
    if (tid == 0) { pending = 0; progress = 0; }
    __syncthreads();
    if (mine) {
      bool kill = false, wait = false;
      for (int dy = -dist; dy <= dist && !kill; ++dy) {
        const int row = c + dy * P;
        for (int dx = -dist; dx <= dist; ++dx) {
          const uint8_t s = ss[row + dx];
          if (s == ST_EMPTY || (dy == 0 && dx == 0)) continue;
          if (s == ST_KEPT) { kill = true; break; }
          const float vj = sv[row + dx];
          if (vj > v || (vj == v && (dy < 0 || (dy == 0 && dx < 0)))) wait = true;  // lower row-major index wins ties
        }
      }
      if (kill || !wait) {
        const uint8_t ns = kill ? ST_EMPTY : ST_KEPT;
        ss[c] = ns;
        st_state(w.state + gi, ns);
        if (!kill) push_kept(w, gi, gx, gy, v, H, W, border, cap2);
        mine = false;
        progress = 1;
      } else {
        pending = 1;
      }
    }
    __syncthreads();
    if (!pending) break;
    for (int k = tid; k < P * P; k += 1024) {
      const int py = k / P, px = k % P;
      if (py >= dist && py < dist + NMS_TILE && px >= dist && px < dist + NMS_TILE) continue;  // interior: ours
      if (ss[k] != ST_UNDECIDED) continue;
      const uint8_t s = ld_state(w.state + (ty0 - dist + py) * W + tx0 - dist + px);
      if (s != ST_UNDECIDED) { ss[k] = s; progress = 1; }
    }
    __syncthreads();
    if (progress) idle = 0;
    else if (++idle > max_idle) break;
  }
}

// One block of 1024 threads per heatmap: finishes whatever the tiled rounds left undecided (everything when
// dist > NMS_MAX_HALO), then border removal, descending bitonic sort, soft-argmax refinement and top-k.
// pts: [cap][5] = (x, y, confidence, soft-argmax x in [0,4], soft-argmax y in [0,4]); the host adds (sx - 2, sy - 2)
// in float64 like models/model_wrap.py:245.  count: number of rows written.
__global__ __launch_bounds__(1024) void nms_points_kernel(const float* __restrict__ heat, PointsWork w, int H, int W,
                                                          int dist, int border, int top_k, int subpixel, int cap,
                                                          int cap2, float* __restrict__ pts, int32_t* __restrict__ count) {
  __shared__ int n_next;
  const int tid = threadIdx.x;
  if (tid == 0) n_next = 0;
  __syncthreads();
  for (int i = tid; i < H * W; i += 1024)
    if (ld_state(w.state + i) == ST_UNDECIDED) w.cand[0][atomicAdd(&n_next, 1)] = i;
  __syncthreads();
  int n = n_next;
  int cur = 0;
  __syncthreads();
  while (n > 0) {
    if (tid == 0) n_next = 0;
    __syncthreads();
    const int32_t* cl = w.cand[cur];
    int32_t* nl = w.cand[cur ^ 1];
    for (int k = tid; k < n; k += 1024) {
      const int i = cl[k];
      const int y = i / W, x = i - y * W;
      const float v = heat[i];
      bool kill = false, wait = false;
      const int y0 = max(y - dist, 0), y1 = min(y + dist, H - 1), x0 = max(x - dist, 0), x1 = min(x + dist, W - 1);
      for (int yy = y0; yy <= y1 && !kill; ++yy) {
        for (int xx = x0; xx <= x1; ++xx) {
          const int j = yy * W + xx;
          const uint8_t s = ld_state(w.state + j);
          if (s == ST_EMPTY || j == i) continue;
          if (s == ST_KEPT) { kill = true; break; }
          const float vj = heat[j];
          if (vj > v || (vj == v && j < i)) wait = true;
        }
      }
      if (kill) {
        st_state(w.state + i, ST_EMPTY);
      } else if (!wait) {
        st_state(w.state + i, ST_KEPT);
        push_kept(w, i, x, y, v, H, W, border, cap2);
      } else {
        nl[atomicAdd(&n_next, 1)] = i;
      }
    }
    __syncthreads();
    n = n_next;
    cur ^= 1;
    __syncthreads();
  }
  __threadfence();
  const int n_kept = __hip_atomic_load(w.counters + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // descending sort: confidence (positive floats order like their bit patterns), then ascending pixel index
  int p2 = 1;
  const int nk = min(n_kept, cap2);
  while (p2 < nk) p2 <<= 1;
  uint64_t* key = w.keys;
  for (int k2 = 2; k2 <= p2; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < p2; i += 1024) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const bool down = (i & k2) == 0;
          const uint64_t a = key[i], b = key[ixj];
          if ((a < b) == down) {
            key[i] = b;
            key[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  int nout = min(nk, cap);
  if (top_k > 0) nout = min(nout, top_k);
  for (int r = tid; r < nout; r += 1024) {
    const uint64_t k = key[r];
    const int i = (int)(0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull));
    const int y = i / W, x = i - y * W;
    float sx = 2.f, sy = 2.f;
    if (subpixel) soft_argmax5(heat, H, W, x, y, sx, sy);
    float* o = pts + (size_t)r * 5;
    o[0] = (float)x; o[1] = (float)y; o[2] = heat[i]; o[3] = sx; o[4] = sy;
  }
  if (tid == 0) *count = nout;
}

// Logging branch of train_val_sample (Train_model_heatmap_all.py:447-568): heatmap_nms (:693-707) turns the kept
// points into a 0/1 map, precisionRecall_torch (utils/utils.py:929-941) compares it with the label map:
//   precision = sum(pred * labels) / (sum(pred) + 1e-6), recall = sum(pred * labels) / (sum(labels) + 1e-6).
// One block per image.  nms_map (optional) must be zero-filled; pts/count come from nms_points_kernel.
__global__ __launch_bounds__(256) void nms_map_pr_kernel(const float* __restrict__ pts, const int32_t* __restrict__ count,
                                                         const float* __restrict__ labels, float* __restrict__ nms_map,
                                                         float* __restrict__ pr, int H, int W) {
  __shared__ float red[2][4];
  const int tid = threadIdx.x, n = *count;
  float tp = 0.f, nl = 0.f;
  for (int r = tid; r < n; r += 256) {
    const int x = (int)pts[r * 5], y = (int)pts[r * 5 + 1];
    if (nms_map) nms_map[y * W + x] = 1.f;
    if (labels) tp += labels[y * W + x];
  }
  if (labels)
    for (int i = tid; i < H * W; i += 256) nl += labels[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    tp += __shfl_xor(tp, o);
    nl += __shfl_xor(nl, o);
  }
  if ((tid & 63) == 0) { red[0][tid >> 6] = tp; red[1][tid >> 6] = nl; }
  __syncthreads();
  if (tid == 0 && pr) {
    tp = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    nl = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    pr[0] = tp / ((float)n + 1e-6f);
    pr[1] = tp / (nl + 1e-6f);
  }
}

}  // namespace sspk
