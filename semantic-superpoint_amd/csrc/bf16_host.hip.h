// Host side of the bf16 path (conv algorithm 12): launchers of conv_bf16.hip.h / wgrad_bf16.hip.h and their operator-level
// entry points.  Included by ssp.hip after its helpers (fail, CHK, HIPCHK, cdiv, AttrOnce, device_cu_count).
#pragma once

struct ConvBCall {
  const void* in[2] = {nullptr, nullptr}; int in_cs = 0, in_co = 0, cin = 0;
  const uint16_t* wpk = nullptr; const float* bias = nullptr;
  void* out[2] = {nullptr, nullptr}; int out_cs = 0, out_co = 0, cout = 0;
  const float* in_scale[2] = {nullptr, nullptr}; const float* in_shift[2] = {nullptr, nullptr};
  double* stats[2] = {nullptr, nullptr};
  uint16_t* pool_out[2] = {nullptr, nullptr}; const float* pool_gamma = nullptr;
  const uint16_t* bnr_t[2] = {nullptr, nullptr};   // fused BatchNorm-backward sums of the layer below (ConvBArgs::bnr_*)
  const float* bnr_scale[2] = {nullptr, nullptr}; const float* bnr_shift[2] = {nullptr, nullptr};
  const float* bnr_mean[2] = {nullptr, nullptr}; const float* bnr_invstd[2] = {nullptr, nullptr};
  double* bnr_sums[2] = {nullptr, nullptr};
  int nviews = 1, N = 0, H = 0, W = 0, ks = 3, in_mode = 0;
  bool in_f32 = false, out_f32 = false;
};

static inline int convb_nchunks(int cin) { return cdiv(cin, CB_KC); }
static inline int convb_ncob(int cout) { return cdiv(cout, CB_NB); }
static inline size_t convb_image_bytes(int ks, int cin, int cout) {
  return (size_t)convb_ncob(cout) * convb_nchunks(cin) * ks * ks * 4096;
}

template <int KS, int IN_MODE, bool IN_F32, bool OUT_F32>
static int launch_conv_bf16_t(const ConvBArgs& a, int nblocks, hipStream_t st) {
  using G = ConvBGeom<KS>;
  static AttrOnce attr_once;
  auto kern = conv_bf16_kernel<KS, IN_MODE, IN_F32, OUT_F32>;
  if (attr_once.need())
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), G::LDS_BYTES, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

template <int IN_MODE, bool NC2>
static int launch_conv_bf16_ws_t(const ConvBArgs& a, int nblocks, hipStream_t st) {
  static AttrOnce attr_once;
  auto kern = conv_bf16_ws_kernel<IN_MODE, NC2>;
  if (attr_once.need())
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ConvWsGeom::LDS_BYTES));
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(conv_ws_threads<IN_MODE>()), ConvWsGeom::LDS_BYTES, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

// whether launch_conv_bf16 runs this call on the wave-specialised kernel (the form that can carry the fused BatchNorm-backward sums)
static bool launch_conv_bf16_is_ws(const ConvBCall& c) {
  static const int ws_env = getenv("SSP_CONVB_WS") ? atoi(getenv("SSP_CONVB_WS")) : 1;
  const int nchunks = convb_nchunks(c.cin);
  return ws_env != 0 && c.ks == 3 && !c.in_f32 && !c.out_f32 && nchunks >= 2 && nchunks % 2 == 0 && c.cin % CB_KC == 0 &&
         c.in_mode == 0 && c.bias == nullptr && c.out_co == 0 && c.out_cs == c.cout && !c.stats[0];
}

static int launch_conv_bf16(const ConvBCall& c, int n_cu, hipStream_t st) {
  ConvBArgs a;
  for (int k = 0; k < 2; ++k) {
    a.in[k] = c.in[k]; a.out[k] = c.out[k]; a.in_scale[k] = c.in_scale[k]; a.in_shift[k] = c.in_shift[k];
    a.stats[k] = c.stats[k]; a.pool_out[k] = c.pool_out[k];
    a.bnr_t[k] = c.bnr_t[k]; a.bnr_scale[k] = c.bnr_scale[k]; a.bnr_shift[k] = c.bnr_shift[k]; a.bnr_mean[k] = c.bnr_mean[k];
    a.bnr_invstd[k] = c.bnr_invstd[k]; a.bnr_sums[k] = c.bnr_sums[k];
  }
  a.wpk = c.wpk; a.bias = c.bias; a.pool_gamma = c.pool_gamma;
  a.nviews = c.nviews; a.N = c.N; a.H = c.H; a.W = c.W;
  a.Cin = c.cin; a.in_cs = c.in_cs; a.in_co = c.in_co; a.Cout = c.cout; a.out_cs = c.out_cs; a.out_co = c.out_co;
  a.tiles_x = cdiv(c.W, CB_T); a.tiles_y = cdiv(c.H, CB_T);
  a.nchunks = convb_nchunks(c.cin); a.ncob = convb_ncob(c.cout);
  const double img = (double)c.H * c.W * c.in_cs * (c.in_f32 ? 4.0 : 2.0);
  if (img > 2147483647.0) return fail(-3, "bf16 conv: one input image [%d,%d,%d] exceeds 2 GiB", c.H, c.W, c.in_cs);
  a.in_img_bytes = (unsigned)img;
  static const int ablate_env = getenv("SSP_CONVB_ABLATE") ? atoi(getenv("SSP_CONVB_ABLATE")) : 0;  // (perf-debug)
  a.ablate = ablate_env;
  a.trace = nullptr;
  static const int trace_env = getenv("SSP_CONVB_TRACE") ? atoi(getenv("SSP_CONVB_TRACE")) : 0;  // (perf-debug: blocking, prints per launch)
  static unsigned long long* trace_buf = nullptr;
  static int trace_count = 0;
  const bool trace_now = trace_env > 0 && c.ks == 3 && !c.in_f32 && !c.out_f32 && (++trace_count % trace_env) == 0;   // every N-th launch
  if (trace_now) {
    if (!trace_buf) HIPCHK(hipMalloc(&trace_buf, (64 + 1024) * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(trace_buf, 0, (64 + 1024) * sizeof(unsigned long long), st));
    a.trace = trace_buf;
  }
  if (!c.out_f32 && (c.cout % 8 || c.out_cs % 8 || c.out_co % 8)) return fail(-3, "bf16 conv: bf16 output needs channel counts / offsets that are multiples of 8");
  if (!c.in_f32 && (c.in_cs % 8 || c.in_co % 8)) return fail(-3, "bf16 conv: bf16 input needs channel stride / offset that are multiples of 8");
  if (c.in_f32 && (c.in_cs % 4 || c.in_co % 4)) return fail(-3, "bf16 conv: fp32 input needs channel stride / offset that are multiples of 4");
  if (c.in_mode == 1 && (!c.in_scale[0] || !c.in_shift[0])) return fail(-1, "bf16 conv: in_mode 1 needs scale / shift");
  if (c.pool_out[0] && (c.out_f32 || c.H % 2 || c.W % 2 || c.out_co != 0 || c.out_cs != c.cout || !c.pool_gamma))
    return fail(-3, "bf16 conv: pooled raw output needs a dense bf16 output of even size");
  const long units = (long)c.nviews * a.ncob * c.N * a.tiles_x * a.tiles_y;
  int nblocks = (int)std::min<long>(2L * n_cu, cdiv(units, 8) * 8L) / 8 * 8;
  nblocks = std::max(nblocks, 8);
  static const int grid_env = getenv("SSP_CONVB_GRID") ? atoi(getenv("SSP_CONVB_GRID")) : 0;  // (perf-debug)
  if (grid_env > 0) nblocks = grid_env;
  // the 3x3 layers with bf16 tensors at both ends: wave-specialised kernel, one 8-wave workgroup per CU (SSP_CONVB_WS=0: perf-debug A/B)
  static const int ws_env = getenv("SSP_CONVB_WS") ? atoi(getenv("SSP_CONVB_WS")) : 1;
  const bool ws_ok = ws_env != 0 && c.ks == 3 && !c.in_f32 && !c.out_f32 && a.nchunks >= 2 && a.nchunks % 2 == 0 && c.cin % CB_KC == 0 &&
                     !(c.in_mode == 0 && c.bias != nullptr);   // (the staging-free form is the data gradient: no bias path)
  // fused BatchNorm-backward sums: the wave-specialised data-gradient form with a dense output, or the generic kernel with a bf16 output
  // (there the tensor of the layer below has the geometry of the output: out_cs, out_co) - never beside forward statistics
  if (c.bnr_t[0] != nullptr && (c.stats[0] || c.out_f32 || (ws_ok && !(c.in_mode == 0 && c.out_co == 0 && c.out_cs == c.cout))))
    return fail(-3, "bf16 conv: the fused BatchNorm-backward sums need a bf16 output without forward statistics (wave-specialised form: dense output)");
  if (ws_ok) {
    int nb = (int)std::min<long>((long)n_cu, cdiv(units, 8) * 8L) / 8 * 8;
    nb = std::max(nb, 8);
    if (grid_env > 0) nb = grid_env;
    const bool nc2 = a.nchunks == 2;   // a layer of 64 input channels: its affine stays in registers
    const int rc = c.in_mode == 1 ? (nc2 ? launch_conv_bf16_ws_t<1, true>(a, nb, st) : launch_conv_bf16_ws_t<1, false>(a, nb, st))
                                  : (nc2 ? launch_conv_bf16_ws_t<0, true>(a, nb, st) : launch_conv_bf16_ws_t<0, false>(a, nb, st));
    if (trace_now && rc == 0) {
      static unsigned long long h[64 + 1024];
      HIPCHK(hipStreamSynchronize(st));
      HIPCHK(hipMemcpy(h, trace_buf, sizeof(h), hipMemcpyDeviceToHost));
      fprintf(stderr, "[convb trace] %dx%d cin %d cout %d mode %d: %llu stages of workgroup 0; cycles per stage\n", c.H, c.W, c.cin, c.cout, c.in_mode, h[7]);
      const double ns = (double)std::max<unsigned long long>(h[7], 1);
      {
        unsigned long long first = ~0ull, last = 0; double sum = 0, mx = 0, mn = 1e30; int cnt = 0;
        for (int b = 0; b < std::min(nb, 512); ++b) if (h[65 + 2 * b]) { first = std::min(first, h[64 + 2 * b]); last = std::max(last, h[65 + 2 * b]); }
        double xs[8] = {0}, xe[8] = {0}; int xn[8] = {0};
        for (int b = 0; b < std::min(nb, 512); ++b) if (h[65 + 2 * b]) {
          const double d = (h[65 + 2 * b] - h[64 + 2 * b]) / 100.0; sum += d; mx = std::max(mx, d); mn = std::min(mn, d); ++cnt;
          xs[b & 7] += (h[64 + 2 * b] - first) / 100.0; xe[b & 7] += (h[65 + 2 * b] - first) / 100.0; ++xn[b & 7];
        }
        fprintf(stderr, "  workgroups: %d, loop us min %.1f avg %.1f max %.1f; first start -> last end %.1f us\n", cnt, mn, sum / std::max(cnt, 1), mx, (last - first) / 100.0);
        for (int x = 0; x < 8; ++x) if (xn[x]) fprintf(stderr, "    xcd %d: avg start %.1f  avg end %.1f us\n", x, xs[x] / xn[x], xe[x] / xn[x]);
      }
      fprintf(stderr, "  loop: %llu cycles in %.1f us -> shader clock %.0f MHz\n", h[5], h[6] / 100.0, h[6] ? h[5] * 100.0 / h[6] : 0.0);
      for (int w = 0; w < 4; ++w) fprintf(stderr, "  consumer %d: mfma %.0f  rest %.0f  barrier %.0f\n", w, h[w * 8] / ns, h[w * 8 + 1] / ns, h[w * 8 + 2] / ns);
      for (int w = 4; w < 8; ++w) fprintf(stderr, "  producer %d: copy-out %.0f  load wait %.0f  staging %.0f  advance %.0f  issue %.0f  barrier %.0f\n", w - 4, h[w * 8] / ns, h[w * 8 + 4] / ns, h[w * 8 + 1] / ns, h[w * 8 + 5] / ns, h[w * 8 + 2] / ns, h[w * 8 + 3] / ns);
    }
    return rc;
  }
#define CONVB_CASE(KS_, M_, I_, O_) \
  if (c.ks == KS_ && c.in_mode == M_ && c.in_f32 == I_ && c.out_f32 == O_) return launch_conv_bf16_t<KS_, M_, I_, O_>(a, nblocks, st);
  CONVB_CASE(3, 1, false, false) CONVB_CASE(3, 0, false, false) CONVB_CASE(1, 1, false, true) CONVB_CASE(1, 0, true, false)
#undef CONVB_CASE
  return fail(-3, "bf16 conv: unsupported variant ks=%d in_mode=%d in_f32=%d out_f32=%d", c.ks, c.in_mode, (int)c.in_f32, (int)c.out_f32);
}

static int launch_pack_bf16(const float* w, uint16_t* dst, int cout_w, int cin_w, int ks, int tf, int nchunks_total, int chunk_off,
                            hipStream_t st) {
  const int conv_cin = tf ? cout_w : cin_w, conv_cout = tf ? cin_w : cout_w;
  const int nchunks = convb_nchunks(conv_cin), ncob = convb_ncob(conv_cout);
  const long total = (long)ncob * nchunks * ks * ks * 2048;
  hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, dst, cout_w, cin_w, ks, tf,
                     nchunks_total > 0 ? nchunks_total : nchunks, chunk_off, ncob, nchunks);
  HIPCHK(hipGetLastError());
  return 0;
}

// job table of pack_weights_bf16_multi_kernel: add() queues an image, flush() launches what is queued
struct PackBQueue {
  PackBJobs J;
  int nblocks;
  PackBQueue() : nblocks(0) { J.n = 0; }
  int flush(hipStream_t st) {
    if (J.n == 0) return 0;
    hipLaunchKernelGGL(pack_weights_bf16_multi_kernel, dim3(nblocks), dim3(256), 0, st, J);
    HIPCHK(hipGetLastError());
    J.n = 0; nblocks = 0;
    return 0;
  }
  int add(const float* w, uint16_t* dst, int cout_w, int cin_w, int ks, int tf, int nchunks_total, int chunk_off, hipStream_t st) {
    if (J.n == PACKB_MAX_JOBS) CHK(flush(st));
    const int conv_cin = tf ? cout_w : cin_w, conv_cout = tf ? cin_w : cout_w;
    PackBJob& q = J.j[J.n++];
    q.w = w; q.dst = dst; q.cout_w = cout_w; q.cin_w = cin_w; q.ks = ks; q.tf = tf;
    q.nchunks = convb_nchunks(conv_cin); q.ncob = convb_ncob(conv_cout);
    q.nchunks_total = nchunks_total > 0 ? nchunks_total : q.nchunks; q.chunk_off = chunk_off;
    q.block0 = nblocks;
    nblocks += (int)cdiv((long)q.ncob * q.nchunks * ks * ks * 2048, 256);
    return 0;
  }
};

struct WgradBCall {
  const void* x[2] = {nullptr, nullptr}; int x_cs = 0, x_co = 0, cin = 0;
  const void* dy[2] = {nullptr, nullptr}; int dy_cs = 0, dy_co = 0, cout = 0;
  const float* x_scale[2] = {nullptr, nullptr}; const float* x_shift[2] = {nullptr, nullptr};
  float* dw = nullptr;  // OIHW gradient, accumulated
  int nviews = 1, N = 0, H = 0, W = 0, ks = 3, in_mode = 0;
  bool dy_f32 = false;
  // fuse 1 / 2: the APPLY pass of this layer's BatchNorm + ReLU (+ MaxPool) backward rides the dY staging (WgradBArgs::f_*); dy = the
  // tensor the kernel WRITES dY to (for the data gradient), f_y / f_dout what it reads
  int fuse = 0;
  const void* f_y[2] = {nullptr, nullptr}; const void* f_dout[2] = {nullptr, nullptr};
  const float* f_scale[2] = {nullptr, nullptr}; const float* f_shift[2] = {nullptr, nullptr}; const float* f_mean[2] = {nullptr, nullptr};
  const float* f_invstd[2] = {nullptr, nullptr}; const float* f_k12[2] = {nullptr, nullptr}; const float* f_gamma = nullptr;
  int f_dcs = 0, f_dco = 0;
};

template <int KS, int IN_MODE, bool DY_F32, int FUSE = 0>
static int launch_wgrad_bf16_t(const WgradBArgs& a, int nblocks, hipStream_t st) {
  using G = WgradBGeom<KS>;
  static AttrOnce attr_once;
  auto kern = wgrad_bf16_kernel<KS, IN_MODE, DY_F32, FUSE>;
  constexpr int lds = G::LDS_BYTES + (FUSE != 0 ? G::P_BYTES : 0);
  static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
  if (attr_once.need())
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

// Deferred slab reductions of a backward pass (the engine's launches): every weight gradient keeps its own slice of the partial
// buffer until flush() sums them all in ONE launch (wgrad_reduce_multi_kernel).  Two jobs never write the same gradient.
struct WredBQueue {
  WredBJobs J;
  size_t used;   // floats of the partial buffer taken by the queued jobs
  WredBQueue() : used(0) { J.n = 0; }
  int flush(hipStream_t st) {
    if (J.n == 0) { used = 0; return 0; }
    const WredBJob& last = J.j[J.n - 1];
    const int nblocks = last.block0 + cdiv(last.cout * last.cin * last.ks * last.ks, 64);
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(nblocks), dim3(256), 0, st, J);
    HIPCHK(hipGetLastError());
    J.n = 0; used = 0;
    return 0;
  }
};

// partial slabs [pairs * nsplit][taps][64][64] in `partial`, summed into c.dw by wgrad_reduce_kernel (at once, or - with a queue -
// by the queue's flush)
static int launch_wgrad_bf16(const WgradBCall& c, float* partial, size_t partial_floats, int n_cu, hipStream_t st,
                             WredBQueue* queue = nullptr) {
  WgradBArgs a;
  for (int k = 0; k < 2; ++k) { a.x[k] = c.x[k]; a.dy[k] = c.dy[k]; a.x_scale[k] = c.x_scale[k]; a.x_shift[k] = c.x_shift[k]; }
  a.partial = partial; a.nviews = c.nviews; a.N = c.N; a.H = c.H; a.W = c.W;
  a.Cin = c.cin; a.x_cs = c.x_cs; a.x_co = c.x_co; a.Cout = c.cout; a.dy_cs = c.dy_cs; a.dy_co = c.dy_co;
  a.tiles_x = cdiv(c.W, CB_T); a.tiles_y = cdiv(c.H, CB_T);
  a.ncib = cdiv(c.cin, 64); a.ncob = cdiv(c.cout, 64);
  const double ximg = (double)c.H * c.W * c.x_cs * 2.0, dimg = (double)c.H * c.W * c.dy_cs * (c.dy_f32 ? 4.0 : 2.0);
  if (ximg > 2147483647.0 || dimg > 2147483647.0) return fail(-3, "bf16 wgrad: one image exceeds 2 GiB");
  a.x_img_bytes = (unsigned)ximg; a.dy_img_bytes = (unsigned)dimg;
  if (c.x_cs % 8 || c.x_co % 8) return fail(-3, "bf16 wgrad: input channel stride / offset must be multiples of 8");
  if (c.dy_cs % (c.dy_f32 ? 4 : 8) || c.dy_co % (c.dy_f32 ? 4 : 8)) return fail(-3, "bf16 wgrad: dY channel stride / offset misaligned");
  if (c.in_mode == 1 && (!c.x_scale[0] || !c.x_shift[0])) return fail(-1, "bf16 wgrad: in_mode 1 needs scale / shift");
  const int pairs = a.ncib * a.ncob, taps = c.ks * c.ks;
  const long ntiles = (long)c.nviews * c.N * a.tiles_x * a.tiles_y;
  static const int wg_per_cu = getenv("SSP_WGB_PER_CU") ? std::max(1, atoi(getenv("SSP_WGB_PER_CU"))) : 2;  // (perf-debug: slabs per CU)
  long nsplit = std::max(1L, std::min<long>(ntiles, ((long)wg_per_cu * n_cu) / pairs));
  while (nsplit > 1 && (size_t)pairs * nsplit * taps * 4096 > partial_floats) --nsplit;
  if ((size_t)pairs * nsplit * taps * 4096 > partial_floats) return fail(-4, "bf16 wgrad: scratch too small");
  a.nsplit = (int)nsplit;
  if (queue != nullptr) {   // own slice of the buffer; the reduction waits for the flush
    const size_t need = (size_t)pairs * nsplit * taps * 4096;
    if (queue->used + need > partial_floats || queue->J.n == WREDB_MAX_JOBS) CHK(queue->flush(st));
    a.partial = partial + queue->used;
    WredBJob& q = queue->J.j[queue->J.n];
    q.partial = a.partial; q.dw = c.dw; q.cin = c.cin; q.cout = c.cout; q.ks = c.ks; q.ncob = a.ncob; q.nsplit = (int)nsplit;
    q.block0 = queue->J.n ? queue->J.j[queue->J.n - 1].block0 + cdiv(queue->J.j[queue->J.n - 1].cout * queue->J.j[queue->J.n - 1].cin *
                                                                     queue->J.j[queue->J.n - 1].ks * queue->J.j[queue->J.n - 1].ks, 64) : 0;
    ++queue->J.n;
    queue->used += need;
  }
  const int nblocks = pairs * (int)nsplit;
  if (c.fuse != 0) {
    if (c.ks != 3 || c.dy_f32 || (c.fuse != 1 && c.fuse != 2) || !c.f_y[0] || !c.f_dout[0] || !c.f_gamma || !c.f_k12[0])
      return fail(-3, "bf16 wgrad: the fused BatchNorm-backward APPLY needs a 3x3 layer with bf16 dY and its y / dOut / parameters");
    if (c.fuse == 2 && ((c.H | c.W) & 1)) return fail(-3, "bf16 wgrad: fused pooled APPLY on an odd map");
    if (c.f_dcs % 8 || c.f_dco % 8) return fail(-3, "bf16 wgrad: dOut channel stride / offset must be multiples of 8");
    for (int k = 0; k < 2; ++k) {
      a.f_y[k] = c.f_y[k]; a.f_dout[k] = c.f_dout[k]; a.f_dy[k] = const_cast<void*>(c.dy[k]); a.f_scale[k] = c.f_scale[k];
      a.f_shift[k] = c.f_shift[k]; a.f_mean[k] = c.f_mean[k]; a.f_invstd[k] = c.f_invstd[k]; a.f_k12[k] = c.f_k12[k];
    }
    a.f_gamma = c.f_gamma; a.f_dcs = c.f_dcs; a.f_dco = c.f_dco;
  }
#define WGB_CASE(KS_, M_, F_) \
  if (c.ks == KS_ && c.in_mode == M_ && c.dy_f32 == F_ && c.fuse == 0) { CHK((launch_wgrad_bf16_t<KS_, M_, F_>(a, nblocks, st))); } else
#define WGB_FCASE(M_, FU_) \
  if (c.ks == 3 && c.in_mode == M_ && !c.dy_f32 && c.fuse == FU_) { CHK((launch_wgrad_bf16_t<3, M_, false, FU_>(a, nblocks, st))); } else
  WGB_FCASE(1, 1) WGB_FCASE(1, 2) WGB_FCASE(0, 1) WGB_FCASE(0, 2)
  WGB_CASE(3, 1, false) WGB_CASE(3, 0, false) WGB_CASE(1, 1, true)
  return fail(-3, "bf16 wgrad: unsupported variant ks=%d in_mode=%d dy_f32=%d", c.ks, c.in_mode, (int)c.dy_f32);
#undef WGB_CASE
#undef WGB_FCASE
  if (queue != nullptr) return 0;
  const int total = c.cout * c.cin * taps;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 64)), dim3(256), 0, st, partial, c.dw, c.cin, c.cout, c.ks, a.ncob, a.nsplit);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" {

// Operator-level entry of wgrad_bf16_kernel: x bf16 NHWC [n,h,w,cin], dy bf16 (fp32 when dy_f32) NHWC [n,h,w,cout]; the OIHW
// fp32 gradient is ACCUMULATED into dw_oihw_dev; workspace: partial slabs (>= ceil(cin/64) ceil(cout/64) ksize^2 16 KiB).
int ssp_op_conv_wgrad_bf16(const void* x_dev, const void* dy_dev, float* dw_oihw_dev, int n, int hh, int w, int cin, int cout,
                           int ksize, int in_mode, const float* in_scale_dev, const float* in_shift_dev, int dy_f32,
                           void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (ksize != 1 && ksize != 3) return fail(-1, "bf16 wgrad: ksize must be 1 or 3");
  WgradBCall c;
  c.x[0] = x_dev; c.x_cs = cin; c.cin = cin; c.dy[0] = dy_dev; c.dy_cs = cout; c.cout = cout;
  c.x_scale[0] = in_scale_dev; c.x_shift[0] = in_shift_dev; c.dw = dw_oihw_dev; c.N = n; c.H = hh; c.W = w; c.ks = ksize;
  c.in_mode = in_mode; c.dy_f32 = dy_f32 != 0;
  return launch_wgrad_bf16(c, reinterpret_cast<float*>(workspace_dev), workspace_bytes / sizeof(float), device_cu_count(),
                           (hipStream_t)stream);
}

// Operator-level entry of conv_bf16_kernel (unit parity tests).  in: bf16 (or fp32 when in_f32) NHWC [n,h,w,cin]; weights OIHW
// fp32 (rounded to bf16 by the packing kernel); out: bf16 (or fp32 when out_f32) NHWC [n,h,w,cout]; pool_out (optional, bf16
// [n,h/2,w/2,cout]) with pool_gamma [cout]; workspace: >= ceil(cout/64) ceil(cin/32) ksize^2 4096 bytes.
int ssp_op_conv_bf16(const void* in_dev, const float* w_oihw_dev, const float* bias_dev, void* out_dev, int n, int hh, int w, int cin,
                     int cout, int ksize, int in_mode, const float* in_scale_dev, const float* in_shift_dev, double* stats_dev,
                     int transpose_flip, int in_f32, int out_f32, void* pool_out_dev, const float* pool_gamma_dev,
                     void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (ksize != 1 && ksize != 3) return fail(-1, "bf16 conv: ksize must be 1 or 3");
  const size_t need = convb_image_bytes(ksize, cin, cout);
  if (workspace_bytes < need) return fail(-4, "ssp_op_conv_bf16 workspace too small (%zu < %zu)", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  uint16_t* wpk = reinterpret_cast<uint16_t*>(workspace_dev);
  if (!transpose_flip) CHK(launch_pack_bf16(w_oihw_dev, wpk, cout, cin, ksize, 0, 0, 0, st));
  else CHK(launch_pack_bf16(w_oihw_dev, wpk, cin, cout, ksize, 1, 0, 0, st));
  ConvBCall c;
  c.in[0] = in_dev; c.in_cs = cin; c.in_co = 0; c.cin = cin; c.wpk = wpk; c.bias = bias_dev; c.out[0] = out_dev; c.out_cs = cout;
  c.out_co = 0; c.cout = cout; c.in_scale[0] = in_scale_dev; c.in_shift[0] = in_shift_dev; c.stats[0] = stats_dev;
  c.pool_out[0] = reinterpret_cast<uint16_t*>(pool_out_dev); c.pool_gamma = pool_gamma_dev;
  c.N = n; c.H = hh; c.W = w; c.ks = ksize; c.in_mode = in_mode; c.in_f32 = in_f32 != 0; c.out_f32 = out_f32 != 0;
  return launch_conv_bf16(c, device_cu_count(), st);
}

}  // extern "C"
