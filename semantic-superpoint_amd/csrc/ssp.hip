// C-ABI library of the MI355X-native Semantic-SuperPoint pair-training path (see include/ssp_hip.h).
// Host orchestration only: every arithmetic step is a HIP kernel from the three headers below.
#include "../../include/ssp_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "bn_kernels.hip.h"
#include "conv_mfma.hip.h"
#include "conv1x1_group.hip.h"
#include "conv_wino.hip.h"
#include "conv_wino_pipe.hip.h"
#include "conv_wino_p2.hip.h"
#include "conv_wino4.hip.h"
#include "wgrad_wino4.hip.h"
#include "wgrad_wino_fused.hip.h"
#ifndef SSP_LEGACY_ALGOS
#define SSP_LEGACY_ALGOS 0
#endif
#if SSP_LEGACY_ALGOS
#include "conv_wino_bf16.hip.h"
#endif
#include "conv_bf16.hip.h"
#include "conv_bf16_ws.hip.h"
#include "wgrad_bf16.hip.h"
#include "loss_kernels.hip.h"
#include "dense_loss.hip.h"
#include "pair_kernels.hip.h"
#include "export_kernels.hip.h"
#include "sem_kernels.hip.h"
#include "backward_tail.hip.h"

using namespace sspk;

// Convolution algorithms of rounds 1-3 that no shipped configuration selects: 2 (conv_wino_kernel, the un-pipelined F(2x2,3x3)
// convolution), 5 (conv_wino_pipe_kernel with the weights staged through LDS) and the bf16-OPERAND experiments inside the fp32
// Winograd kernels that the bf16 path (algorithm 12) superseded: 3 (one bf16 part), 7 (hi + lo parts), 8 (mixed: fp32 forward,
// bf16-operand gradients) - conv_wino_bf16.hip.h.  Compiled out of the shipped library (28 kernel instances); their results are
// profiles/PERF_LOG_rounds_1-4.md section 10 and profiles/r0[2-4]_*mixed_bf16*; -DSSP_LEGACY_ALGOS=1 (SSP_HIPCC_EXTRA) brings them back.
#ifndef SSP_LEGACY_ALGOS
#define SSP_LEGACY_ALGOS 0
#endif

static thread_local std::string g_err;
static int g_dbg_ablate = 0, g_dbg_grid = 0;  // perf-debug knobs of conv_mfma_kernel (tools/archive/ablate_conv.py)
// ssp_set_conv_algo: 0 = direct implicit GEMM, 1 = Winograd F(2x2,3x3) where eligible (software-pipelined kernel),
// 2 = Winograd, un-pipelined kernel (conv_wino_kernel; kept for A/B measurements),
// 3 = Winograd with bf16 matrix-core operands (conv_wino_bf16_kernel: opt-in reduced precision, BASELINE configs[3])
static int g_default_conv_algo = 1;             // process default: new handles and the handle-less ssp_op_* calls
static thread_local int g_conv_algo = 1;        // algorithm of the call in flight (AlgoScope: the handle's, else the default)
// 3x3 convolutions whose input channels fill whole 16-channel K-chunks run as Winograd F(2x2,3x3)
static inline bool wino_ok(int ks, int conv_cin) { return g_conv_algo != 0 && ks == 3 && conv_cin % CK == 0; }
// bf16 Winograd modes: number of bf16 parts per operand element.  3: one part everywhere; 7: hi + lo everywhere;
// 8 (mixed): the FORWARD convolutions (the activations every later layer and the ReLU gates depend on) run the fp32 default
// algorithm (FwdAlgoScope), the data-gradient and weight-gradient kernels take one part (unbiased 2^-9 noise on the gradients,
// like any bf16 training)
// fp32 pipelined Winograd family: 1 (default: F(4x4,3x3) on the large maps - conv_uses_w4 -, F(2x2,3x3) elsewhere), 9 = F(2x2,3x3)
// only (the default of rounds 1-2), 10 = F(4x4,3x3) wherever legal
// 11 = algorithm 1 with the Winograd F(3x3,4x4) weight gradient (wgrad_wino4_kernel: opt-in, measured not faster)
static inline bool pipe_algo() { return g_conv_algo == 1 || g_conv_algo == 9 || g_conv_algo == 10 || g_conv_algo == 11; }
static inline bool bf16_algo() { return SSP_LEGACY_ALGOS && (g_conv_algo == 3 || g_conv_algo == 7 || g_conv_algo == 8); }
// 12 = the bf16 PATH (BASELINE configs[3]): bf16 NHWC activations / activation gradients in HBM, direct implicit-GEMM 3x3 convolutions,
// data and weight gradients on v_mfma_f32_32x32x16_bf16 (conv_bf16.hip.h, wgrad_bf16.hip.h); fp32 master weights, BatchNorm
// statistics, losses and Adam; the pointwise heads and everything behind them stay on the fp32 kernels
static inline bool bf16_path() { return g_conv_algo == 12; }
static inline int bf16_parts(bool backward) { return g_conv_algo == 7 || (g_conv_algo == 8 && !backward) ? 2 : 1; }
static inline int pk_taps(int ks) { return ks == 3 ? W4C : ks * ks; }  // packed-weight capacity per (chunk, 16 ci, 64 co)
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(x)                                                                          \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) return fail(-2, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define CHK(x)             \
  do {                     \
    int r_ = (x);          \
    if (r_ != 0) return r_; \
  } while (0)

// hipFuncSetAttribute is per device: one flag per (kernel instantiation, device)
struct AttrOnce {
  bool done[64] = {};
  bool need() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if (done[dev]) return false;
    done[dev] = true;
    return true;
  }
};

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// CUs of the current device (handle-less operator entry points; queried once per device)
static int device_cu_count() {
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cached[dev];
}
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ------------------------------------------------------------------------------------------------
struct LayerDesc {
  int cin, cout, ks;
  bool bn;
  size_t w_off, b_off, g_off, be_off;  // offsets into the flat parameter vector
  int bn_index;                        // index among BN layers, -1 if none
  size_t bn_ch_off;                    // channel offset into bn_running
  // packed weights (floats offset into wpk_fwd / wpk_bwd)
  size_t pk_fwd, pk_bwd;
  int nchunks_fwd, ncob_fwd, nchunks_bwd, ncob_bwd;
};

struct BnBufs {
  double* stats;   // [2C]
  double* bsums;   // [2C] backward sums
  float *scale, *shift, *mean, *invstd;
  float* k12;      // [2C] reduced backward sums / n
};

struct Slot {
  float* Y[16];     // raw conv outputs (heads share Yheads via offsets)
  int y_cs[16], y_co[16];
  BnBufs bn[16];
  float* desc;      // [B*cells*256] normalised descriptors
  float* inv_norm;  // [B*cells]
  float* cellmask;  // [B*cells]
  float* dsemi;     // [B*cells*80]
  float* ddesc;     // [B*cells*256]
  float* dsout;     // [B*cells*SOUT_CS] gradient wrt convSout output (ssmall)
  float *gP, *gQ;   // backward ping-pong buffers of this slot (dOut / dY)
  float* Apool[8];  // pooled output of layers l = 1, 3, 5 (inputs of layers 2, 4, 6), else nullptr: the RAW pooled conv output
                    // written by layer l's conv (pool_raw[l], BatchNorm + ReLU applied on load) or maxpool(relu(bn(Y_l)))
  bool pool_raw[8] = {false, false, false, false, false, false, false, false};  // set by this slot's last forward
  // bf16 path: act[l] = bf16(relu(bn(.))) of layer l's (pooled) output at 30x40, l = 5, 6, 7 - the materialised input of the layers
  // whose units would each activate the same halo (layers 6, 7: two 64-channel blocks; the three 3x3 heads: twelve)
  float* act[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool act_valid[8] = {false, false, false, false, false, false, false, false};  // written by this slot's last forward
  const float* x;   // input image of the last forward (caller-owned)
  void* stats_region;
  size_t stats_bytes;
  int N, H, W;
  bool bsums_dirty;
};

struct ssp_handle {
  ssp_config cfg;
  int nlayers, nheads;
  LayerDesc L[16];
  size_t n_params, n_bn_ch;
  int n_bn;
  ssp_buffers buf;
  bool bound;
  size_t ws_bytes;
  Slot slot[2];
  float *wpk_fwd, *wpk_bwd, *wpk_heads_bwd;
  float* partial;    // wgrad partial slabs: every Winograd wgrad launch of a backward pass gets its own slice ...
  size_t partial_floats;
  size_t partial_used = 0;  // ... and the slices are reduced into the gradients by ONE launch (flush_wgrad_reduce)
  struct WredBQueue* rq_bf16 = nullptr;  // the same for the bf16 path's weight gradients (bf16_host.hip.h)
  WredJobs rjobs{};
  bool bsums_fused[16] = {};  // pass 1 of layer l's BatchNorm backward was accumulated by the data-gradient conv above it
  bool sums_lazy[16] = {};    // backward: layer l's bn_bwd_sums_kernel launch was skipped - its fused weight gradient reduces the replicas itself
  bool fin_pending[16] = {};  // forward, training: layer l's statistics are complete but not finalized - the convolution that reads the
  double fin_count[16] = {};  // layer does it in its prologue (BnLazy), or bn_finalize_pending() launches the kernel (count: pixels per view)
  bool apply_fused[16] = {};  // pass 2 (APPLY) of layer l was left to the layer's weight gradient (wgrad_wino_fused_kernel)
  bool bf16_fuse_apply = true;  // bf16 path: the same for wgrad_bf16_kernel<.., FUSE>; SSP_BF16_FUSE_APPLY, read ONCE per backward pass
                                // (bf16_fuse_apply_env: tests switch it between two steps of one process) and part of the graph key
  StepAccum* accum;
  float* dots;       // [B * n_match * n_non] non-match dot products of the current step
  float* dense_coef; // [B * cells * cells] d total / d dot of the dense descriptor loss (cfg.dense_loss), else nullptr
  int sout_cs;
  int conv_algo;     // ssp_handle_set_conv_algo (initialised from the process default of ssp_set_conv_algo)
  // layout of the packed weight images wpk_fwd / wpk_bwd as pack_all last wrote them (they are shared by both slots and
  // re-packed by every forward): the launches of a layer use THIS record, never a re-evaluation of w4_eligible with their own
  // shape (a forward of another shape / view count on the other slot, or ssp_handle_set_conv_algo, between a forward and its
  // backward would otherwise run an F(2x2,3x3) kernel on an F(4x4,3x3) image or the reverse)
  bool pk_w4_fwd[16] = {}, pk_w4_bwd[16] = {};
  // pointwise layers (Pb, Db, Sout): operand images of conv1x1_group_kernel (forward / data gradient), whether pack_all wrote
  // them (else the layer runs conv_mfma_kernel<1, ...> on the wpk_fwd / wpk_bwd images), and the partial slabs of the grouped weight gradient
  float *wpk_g1_fwd[16] = {}, *wpk_g1_bwd[16] = {};
  bool pk_g1[16] = {};
  float* g1_partial = nullptr;  // wgrad1x1_group_kernel: one [256][128] slab per workgroup (2 per CU)
  TailJobs tail = {};           // short reductions queued for the launch at the end of the backward pass (flush_wgrad_reduce)
  int g1_partial_slabs = 0;
  int packed_algo = -1;      // conv algorithm the images were packed for
  bool packed_bwd = false;   // the data-gradient images were packed too
  // captured pair steps (ssp_pair_step_graph): one executable graph per (phase, input signature)
  struct GraphEntry { std::vector<unsigned char> key; hipGraphExec_t exec; };
  std::vector<GraphEntry> graphs;
  uint64_t* graph_seed;  // device word: sampler seed of the captured steps (set by a one-thread kernel before each replay)
  // loss phase of the pair step: the descriptor-loss kernels (L2 gathers / atomics, latency-bound) run on a side stream beside
  // the detector / segmentation-loss kernels (vector-ALU-bound) of the caller's stream; fork / join by events (captured as a
  // fork-join into the hipGraph form).  Created at the first pair step, per device of the handle.
  hipStream_t aux_stream = nullptr;
  hipStream_t aux2_stream = nullptr;   // label-only kernels of the loss phase, beside the first-layer convolution and the weight packing
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_pack_fork = nullptr, ev_pack_join = nullptr;  // the weight images are packed beside the first-layer conv
  hipEvent_t ev_early_fork = nullptr, ev_early_join = nullptr;  // label-only work of the loss phase beside the forward pass
  // profiling
  int prof_family;
  bool prof_paused = false;  // ssp_profile_pause: launches are not bracketed while set (bench.py samples every n-th step)
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used;
  double prof_flops, prof_bytes, prof_exec_flops;
  int64_t prof_launches;
  struct ProfKernel { double flops = 0, exec_flops = 0, bytes = 0; int64_t launches = 0; } prof_k[SSP_PROF_K_COUNT];
  std::vector<unsigned char> ev_kernel;  // kernel bucket of event pair i (ev_pool[2 i], ev_pool[2 i + 1])
  int n_cu;
};

struct AlgoScope {  // makes the handle's conv algorithm the current one for the duration of an entry point
  int prev;
  explicit AlgoScope(const ssp_handle* h) : prev(g_conv_algo) { g_conv_algo = h ? h->conv_algo : g_default_conv_algo; }
  ~AlgoScope() { g_conv_algo = prev; }
};

// Mixed mode 8 = fp32 FORWARD (the default algorithm's kernels: F(4x4,3x3) on the large maps is faster than the split-bf16
// F(2x2,3x3) forward this mode used to run, and exact), bf16 operands in the data-gradient and weight-gradient convolutions.
// Forward launches and the packing of the forward images run under this scope.
struct FwdAlgoScope {
  int prev;
  FwdAlgoScope() : prev(g_conv_algo) { if (prev == 8) g_conv_algo = 1; }
  ~FwdAlgoScope() { g_conv_algo = prev; }
};

enum { L_PA = 8, L_PB = 9, L_DA = 10, L_DB = 11, L_DS = 12, L_SOUT = 13 };
static const int kEnc[8][2] = {{1, 64}, {64, 64}, {64, 64}, {64, 64}, {64, 128}, {128, 128}, {128, 128}, {128, 128}};

static void build_layers(ssp_handle* h) {
  int n = 0;
  auto add = [&](int cin, int cout, int ks, bool bn) {
    LayerDesc& d = h->L[n++];
    d.cin = cin; d.cout = cout; d.ks = ks; d.bn = bn;
  };
  for (int i = 0; i < 8; ++i) add(kEnc[i][0], kEnc[i][1], 3, true);
  add(128, 256, 3, true);   // convPa/bnPa
  add(256, 65, 1, true);    // convPb/bnPb
  add(128, 256, 3, true);   // convDa/bnDa
  add(256, 256, 1, true);   // convDb/bnDb
  h->nheads = 2;
  if (h->cfg.arch == SSP_ARCH_GAUSS2_SSMALL) {
    add(128, 256, 3, true);                  // convDS/bnS1
    add(256, h->cfg.n_classes, 1, false);    // convSout
    h->nheads = 3;
  }
  h->nlayers = n;
  size_t off = 0, bnch = 0, pf = 0, pb = 0;
  int nbn = 0;
  for (int i = 0; i < n; ++i) {
    LayerDesc& d = h->L[i];
    d.w_off = off; off += (size_t)d.cout * d.cin * d.ks * d.ks;
    d.b_off = off; off += d.cout;
    if (d.bn) {
      d.g_off = off; off += d.cout;
      d.be_off = off; off += d.cout;
      d.bn_index = nbn++;
      d.bn_ch_off = bnch; bnch += d.cout;
    } else {
      d.g_off = d.be_off = 0; d.bn_index = -1; d.bn_ch_off = 0;
    }
    const int taps = d.ks * d.ks;
    d.nchunks_fwd = cdiv(d.cin, CK); d.ncob_fwd = cdiv(d.cout, NB);
    d.nchunks_bwd = cdiv(d.cout, CK); d.ncob_bwd = cdiv(d.cin, NB);
    (void)taps;
    d.pk_fwd = pf; pf += (size_t)d.ncob_fwd * d.nchunks_fwd * pk_taps(d.ks) * CK * NB;
    d.pk_bwd = pb; pb += (size_t)d.ncob_bwd * d.nchunks_bwd * pk_taps(d.ks) * CK * NB;
  }
  h->n_params = off; h->n_bn_ch = bnch; h->n_bn = nbn;
}

// resolution (H, W) of layer l's output for an input of H0 x W0
static void layer_res(int l, int H0, int W0, int& H, int& W) {
  int s = 0;
  if (l >= 2) s = 1;
  if (l >= 4) s = 2;
  if (l >= 6) s = 3;
  H = H0 >> s; W = W0 >> s;
}
static int layer_in_mode(int l) { return (l == 2 || l == 4 || l == 6) ? 2 : 1; }

struct Carver {
  char* base; size_t off;
  int nbig = 0;
  template <typename T> T* take(size_t n) {
    off = align_up(off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
  // Activation / gradient tensors: the k-th one starts at k * 72 KiB past a 4 MiB boundary of the ABSOLUTE address.
  // A kernel that streams two tensors whose addresses are congruent modulo a few MiB (e.g. the 600 MiB layer-0/1
  // activations carved back to back) sends its reads and writes to the same HBM channels at the same time: measured
  // 0.998 ms vs 0.835 ms for the 64->64 @240x320 convolution (tools/archive/align_probe.py); any skew >= 8 KiB removes it.
  template <typename T> T* take_skewed(size_t n) {
    constexpr size_t A = (size_t)4 << 20, SK = (size_t)72 << 10;
    const size_t abs0 = reinterpret_cast<size_t>(base) + off;  // base == nullptr (size query): layout relative to 0
    off += align_up(abs0, A) - abs0 + (size_t)(nbig++ % 56) * SK;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

static size_t carve(ssp_handle* h, void* base) {
  Carver c{reinterpret_cast<char*>(base), 0};
  const int B = h->cfg.max_batch, H = h->cfg.height, W = h->cfg.width;
  const size_t cells = (size_t)B * (H / 8) * (W / 8);
  h->sout_cs = (int)align_up(h->cfg.n_classes, 4);
  size_t pf = 0, pb = 0;
  for (int i = 0; i < h->nlayers; ++i) {
    const LayerDesc& d = h->L[i];
    pf = d.pk_fwd + (size_t)d.ncob_fwd * d.nchunks_fwd * pk_taps(d.ks) * CK * NB;
    pb = d.pk_bwd + (size_t)d.ncob_bwd * d.nchunks_bwd * pk_taps(d.ks) * CK * NB;
  }
  h->wpk_fwd = c.take<float>(pf);
  h->wpk_bwd = c.take<float>(pb);
  h->wpk_heads_bwd = c.take<float>((size_t)2 * 16 * h->nheads * WC * CK * NB);
  for (int i = 0; i < h->nlayers; ++i) {
    const LayerDesc& d = h->L[i];
    h->wpk_g1_fwd[i] = h->wpk_g1_bwd[i] = nullptr;
    if (d.ks != 1) continue;
    h->wpk_g1_fwd[i] = c.take<float>((size_t)cdiv(d.cin, G1_KC) * cdiv(d.cout, 32) * G1_TILE_FLOATS);
    h->wpk_g1_bwd[i] = c.take<float>((size_t)cdiv(d.cout, G1_KC) * cdiv(d.cin, 32) * G1_TILE_FLOATS);
  }
  h->g1_partial_slabs = 2 * std::max(h->n_cu, 8);  // launch_g1_wgrad uses at most 2 workgroups per CU
  h->g1_partial = c.take<float>((size_t)h->g1_partial_slabs * G1W_SLAB);
  for (int s = 0; s < 2; ++s) {
    Slot& S = h->slot[s];
    for (int l = 0; l < 8; ++l) {
      int lh, lw; layer_res(l, H, W, lh, lw);
      S.Y[l] = c.take_skewed<float>((size_t)B * lh * lw * h->L[l].cout);
      S.y_cs[l] = h->L[l].cout; S.y_co[l] = 0;
    }
    for (int l = 0; l < 8; ++l) S.Apool[l] = nullptr;
    for (int l = 1; l <= 5; l += 2) {
      int lh, lw; layer_res(l, H, W, lh, lw);
      S.Apool[l] = c.take_skewed<float>((size_t)B * (lh / 2) * (lw / 2) * h->L[l].cout);
    }
    const int hcs = 256 * h->nheads;
    float* yheads = c.take_skewed<float>(cells * hcs);
    S.Y[L_PA] = yheads; S.y_cs[L_PA] = hcs; S.y_co[L_PA] = 0;
    S.Y[L_DA] = yheads; S.y_cs[L_DA] = hcs; S.y_co[L_DA] = 256;
    S.Y[L_PB] = c.take_skewed<float>(cells * 80); S.y_cs[L_PB] = 80; S.y_co[L_PB] = 0;
    S.Y[L_DB] = c.take_skewed<float>(cells * 256); S.y_cs[L_DB] = 256; S.y_co[L_DB] = 0;
    if (h->nheads == 3) {
      S.Y[L_DS] = yheads; S.y_cs[L_DS] = hcs; S.y_co[L_DS] = 512;
      S.Y[L_SOUT] = c.take<float>(cells * h->sout_cs); S.y_cs[L_SOUT] = h->sout_cs; S.y_co[L_SOUT] = 0;
      S.dsout = c.take<float>(cells * h->sout_cs);
    } else {
      S.dsout = nullptr;
    }
    S.desc = c.take_skewed<float>(cells * 256);
    for (int l = 5; l <= 7; ++l) S.act[l] = c.take_skewed<float>(cells * 64);   // (bf16 [cells][128])
    S.inv_norm = c.take<float>(cells);
    S.cellmask = c.take<float>(cells);
    S.dsemi = c.take<float>(cells * 80);
    S.ddesc = c.take_skewed<float>(cells * 256);
    // BN buffers; the fp64 sums of all layers are contiguous so that one memset clears them
    size_t nst = 0;
    for (int l = 0; l < h->nlayers; ++l) nst += 4 * (size_t)h->L[l].cout * NREP;
    double* st = c.take<double>(nst);
    S.stats_region = st; S.stats_bytes = nst * sizeof(double);
    for (int l = 0; l < h->nlayers; ++l) {
      const int C = h->L[l].cout;
      S.bn[l].stats = st; st += 2 * C * NREP;
      S.bn[l].bsums = st; st += 2 * C * NREP;
      S.bn[l].scale = c.take<float>(C); S.bn[l].shift = c.take<float>(C);
      S.bn[l].mean = c.take<float>(C); S.bn[l].invstd = c.take<float>(C);
      S.bn[l].k12 = c.take<float>(2 * C);
    }
  }
  const size_t big = (size_t)B * H * W * 64;
  for (int s = 0; s < 2; ++s) {
    h->slot[s].gP = c.take_skewed<float>(big);
    h->slot[s].gQ = c.take_skewed<float>(big);
  }
  h->partial_floats = (size_t)1024 * 9 * 4096 * 5;  // 10 Winograd launches of 256 blocks x 16 slabs (755 MB)
  h->partial = c.take<float>(h->partial_floats);
  h->accum = c.take<StepAccum>(1);
  h->graph_seed = c.take<uint64_t>(1);
  h->dots = c.take<float>((size_t)B * h->cfg.n_match * h->cfg.n_non);
  {
    const size_t pc = (size_t)(H / 8) * (W / 8);
    h->dense_coef = h->cfg.dense_loss ? c.take_skewed<float>((size_t)std::min(B, 64) * pc * pc) : nullptr;
  }
  // size query (base == nullptr): the skewed tensors are placed relative to the absolute address, so a bound base that
  // is not 4 MiB aligned shifts the whole layout by < 4 MiB
  return align_up(c.off, 256) + (base ? 0 : ((size_t)4 << 20));
}

// ------------------------------------------------------------------------------------------------
// profiling (bench.py roofline leg): hipEvents around the tagged kernel family
// ------------------------------------------------------------------------------------------------
struct ProfScope {
  ssp_handle* h; hipStream_t st; bool on;
  ProfScope(ssp_handle* h_, int family, hipStream_t s, double flops, double bytes, double exec_flops = -1.0,
            int kernel = SSP_PROF_K_OTHER) : h(h_), st(s), on(false) {
    const bool conv = family == SSP_PROF_CONV3X3_FWD || family == SSP_PROF_CONV3X3_DGRAD;
    const bool match = h && family > 0 && !h->prof_paused && (h->prof_family == family || (h->prof_family == SSP_PROF_CONV3X3_ALL && conv) ||
        (h->prof_family == SSP_PROF_CONV3X3_EVERY && (conv || family == SSP_PROF_CONV3X3_WGRAD)));
    if (match && h->ev_used + 2 <= h->ev_pool.size()) {
      on = true;
      (void)hipEventRecord(h->ev_pool[h->ev_used], st);
      const double ex = exec_flops >= 0.0 ? exec_flops : flops;
      h->prof_flops += flops; h->prof_bytes += bytes; h->prof_launches += 1;
      h->prof_exec_flops += ex;
      ssp_handle::ProfKernel& k = h->prof_k[kernel];
      k.flops += flops; k.exec_flops += ex; k.bytes += bytes; k.launches += 1;
      h->ev_kernel[h->ev_used / 2] = (unsigned char)kernel;
    }
  }
  ~ProfScope() {
    if (on) { (void)hipEventRecord(h->ev_pool[h->ev_used + 1], st); h->ev_used += 2; }
  }
};

// ------------------------------------------------------------------------------------------------
// Zero fill on the launch stream as an ordinary kernel.  The pair step clears its accumulation buffers with this instead of
// hipMemsetAsync: inside a captured hipGraph (ssp_pair_step_graph) memset NODES of some buffers were observed not to take
// effect on the second and later replays (ROCm 7.2, depending on the allocation layout of the process: the descriptor- and
// segmentation-gradient buffers kept their previous contents and the atomically accumulated gradients blew up), while
// kernel nodes replay reliably.  Grid-stride 16-byte stores: the few-hundred-KB buffers of the step are latency-bound.
// ------------------------------------------------------------------------------------------------
__global__ void zero_fill_kernel(uint32_t* __restrict__ p0, size_t nwords, uint32_t* __restrict__ p1 = nullptr) {  // (blockIdx.y: range 0 / 1)
  uint32_t* __restrict__ p = blockIdx.y ? p1 : p0;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  if ((reinterpret_cast<size_t>(p) & 15) == 0) {
    uint4* q = reinterpret_cast<uint4*>(p);
    const size_t nq = nwords >> 2;
    for (size_t i = tid; i < nq; i += nth) q[i] = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (nq << 2) + tid; i < nwords; i += nth) p[i] = 0u;
  } else {
    for (size_t i = tid; i < nwords; i += nth) p[i] = 0u;
  }
}
static int dev_zero(void* p, size_t bytes, hipStream_t st) {
  if (bytes == 0) return 0;
  if (bytes % 4 != 0 || (reinterpret_cast<size_t>(p) & 3) != 0) return fail(-1, "dev_zero: unaligned range");
  const size_t nwords = bytes / 4;
  const int nb = (int)std::min<size_t>((nwords / 4 + 255) / 256 + 1, 2048);
  hipLaunchKernelGGL(zero_fill_kernel, dim3(nb), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p), nwords);
  HIPCHK(hipGetLastError());
  return 0;
}
// two ranges of the same size in one launch (the statistics of the two views)
static int dev_zero2(void* p0, void* p1, size_t bytes, hipStream_t st) {
  if (bytes == 0) return 0;
  if (p1 == nullptr || p1 == p0) return dev_zero(p0, bytes, st);
  if (bytes % 4 != 0 || ((reinterpret_cast<size_t>(p0) | reinterpret_cast<size_t>(p1)) & 3) != 0) return fail(-1, "dev_zero2: unaligned range");
  const size_t nwords = bytes / 4;
  const int nb = (int)std::min<size_t>((nwords / 4 + 255) / 256 + 1, 2048);
  hipLaunchKernelGGL(zero_fill_kernel, dim3(nb, 2), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p0), nwords, reinterpret_cast<uint32_t*>(p1));
  HIPCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
template <int KS, int IN_MODE, int SH, int SW>
static int launch_conv_t(const ConvArgs& a, int nblocks, hipStream_t st) {
  using G = ConvGeom<KS, SH, SW>;
  static AttrOnce attr_once;
  auto kern = conv_mfma_kernel<KS, IN_MODE, SH, SW>;
  if (attr_once.need()) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), G::LDS_BYTES, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

struct ConvCall {
  const float* in; int in_cs, in_co, cin;
  const float* wpk; const float* bias;
  float* out; int out_cs, out_co, cout;
  const float* in_scale; const float* in_shift; double* stats;
  int N, H, W, ks, in_mode, nchunks, ncob;
  // optional second problem (same shapes / weights): the other view of the pair
  int nprob = 1;
  const float* in2 = nullptr; float* out2 = nullptr;
  const float* in_scale2 = nullptr; const float* in_shift2 = nullptr; double* stats2 = nullptr;
  bool wino = false;  // wpk holds pack_weights_wino_kernel's image: run conv_wino_kernel
  bool backward = false;  // data-gradient launch (selects the operand precision of the mixed bf16 mode)
  // data-gradient launches: BatchNorm-backward sums of the layer below fused into the epilogue (ConvArgs::bnr_*);
  // honoured by the pipelined Winograd kernel only - can_fuse_bnr() tells the caller
  int bnr_mode = 0;
  const float* bnr_t[2] = {nullptr, nullptr};
  int bnr_cs = 0, bnr_co = 0;
  const float* bnr_p[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
  // forward launches: raw 2x2-pooled copy of the output for a BatchNorm + ReLU + MaxPool consumer (ConvArgs::pool_out);
  // honoured by the pipelined Winograd kernel only - conv_writes_pool() tells the caller
  float* pool_out[2] = {nullptr, nullptr};
  const float* pool_gamma = nullptr;
  bool allow_w4 = true;  // false: wpk is an F(2x2,3x3) image whatever the shape (the concatenated data-gradient weights of the heads)
  int force_w4 = -1;     // engine launches: 1 / 0 = wpk is / is not an F(4x4,3x3) image (ssp_handle::pk_w4_*); -1 = decide from the
                         // shape (operator-level calls, which pack with the same predicate right before the launch)
  BnLazy lazy;           // training forward: the input layer's BatchNorm affine derived by this launch (conv_has_bn_lazy() tells the caller)
};
static bool can_fuse_bnr(const ConvCall& c) {
  return c.wino && (pipe_algo() || g_conv_algo == 5 || g_conv_algo == 6 || bf16_algo()) && c.in_mode == 0 && c.cout % 4 == 0 && c.out_co % 4 == 0 &&
         c.out_cs % 4 == 0;
}

template <int IN_MODE, bool WIDE, bool GB = false>
static int launch_wino_pipe_t(const ConvArgs& a, int nblocks, hipStream_t st) {
  static AttrOnce attr_once;
  auto kern = conv_wino_pipe_kernel<IN_MODE, WIDE, GB>;
  if (attr_once.need()) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PIPE_LDS_BYTES));
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(WINO_THREADS), PIPE_LDS_BYTES, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

template <int IN_MODE, bool WIDE>
static int launch_wino_p2_t(const ConvArgs& a, int nblocks, hipStream_t st) {
  static AttrOnce attr_once;
  auto kern = conv_wino_p2_kernel<IN_MODE, WIDE>;
  if (attr_once.need()) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS_BYTES));
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(P2_THREADS), P2_LDS_BYTES, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

template <int IN_MODE, bool WIDE>
static int launch_wino4_t(const ConvArgs& a_in, int nblocks, hipStream_t st) {
  static AttrOnce attr_once;
  auto kern = conv_wino4_kernel<IN_MODE, WIDE>;
  if (attr_once.need()) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_BYTES));
  }
#if W4_TRACE
  // perf-debug build only: every SSP_W4_TRACE-th launch carries a trace buffer, is waited for and printed (cycles per stage)
  ConvArgs a = a_in;
  static const int trace_env = getenv("SSP_W4_TRACE") ? atoi(getenv("SSP_W4_TRACE")) : 0;
  static unsigned long long* trace_buf = nullptr;
  static int trace_count = 0;
  const bool trace_now = trace_env > 0 && (++trace_count % trace_env) == 0;
  if (trace_now) {
    if (!trace_buf) HIPCHK(hipMalloc(&trace_buf, 64 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(trace_buf, 0, 64 * sizeof(unsigned long long), st));
    a.trace = trace_buf;
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(W4_THREADS), W4_LDS_BYTES, st, a);
  HIPCHK(hipGetLastError());
  if (trace_now) {
    unsigned long long hbuf[64];
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(hbuf, trace_buf, sizeof(hbuf), hipMemcpyDeviceToHost));
    const double ns = (double)std::max<unsigned long long>(hbuf[6], 1);
    fprintf(stderr, "[w4 trace] %dx%d cin %d cout %d mode %d bnr %d nprob %d: %llu stages of workgroup 0; cycles per stage (MFMA floor 4608)\n",
            a.H, a.W, a.Cin, a.Cout, IN_MODE, a.bnr_mode, a.nprob, hbuf[6]);
    for (int w = 0; w < 4; ++w)
      fprintf(stderr, "  wave %d: first half %.0f  barrier A %.0f  second half %.0f  barrier B %.0f  epilogue %.0f  | loop %.0f\n", w,
              hbuf[w * 8] / ns, hbuf[w * 8 + 1] / ns, hbuf[w * 8 + 2] / ns, hbuf[w * 8 + 3] / ns, hbuf[w * 8 + 4] / ns, hbuf[w * 8 + 5] / ns);
  }
#else
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(W4_THREADS), W4_LDS_BYTES, st, a_in);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

#if SSP_LEGACY_ALGOS
template <int IN_MODE, bool WIDE, int NT = 1>
static int launch_wino_bf16_t(const ConvArgs& a, int nblocks, hipStream_t st) {
  static AttrOnce attr_once;
  auto kern = conv_wino_bf16_kernel<IN_MODE, WIDE, NT>;
  constexpr int lds = NT == 1 ? BF16_LDS_BYTES : BF16X2_LDS_BYTES;
  if (attr_once.need()) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(WINO_THREADS), lds, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

template <int IN_MODE, bool WIDE>
static int launch_wino_t(const ConvArgs& a, int nblocks, hipStream_t st) {
  static AttrOnce attr_once;
  auto kern = conv_wino_kernel<IN_MODE, WIDE>;
  if (attr_once.need()) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WINO_LDS_BYTES));
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(WINO_THREADS), WINO_LDS_BYTES, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}
#endif

// default algorithm (1): maps with few first-generation work items per CU (the 30x40 layers: 640 items on 256 CUs =
// 2.5 rounds) run on the finer-grained second-generation kernel (measured 10-15 % faster there, 1-4 % slower on the
// large maps: tools/archive/conv_probe.py)
// Winograd F(4x4,3x3) (conv_wino4_kernel): tile blocks of 32x16 / 16x32 pixels, whichever wastes less of the map
static void w4_geometry(int H, int W, bool& wide, int& tiles_y, int& tiles_x) {
  const long a_w = (long)cdiv(H, 16) * 16 * cdiv(W, 32) * 32, a_t = (long)cdiv(H, 32) * 32 * cdiv(W, 16) * 16;
  wide = a_w <= a_t;
  tiles_y = cdiv(H, wide ? 16 : 32); tiles_x = cdiv(W, wide ? 32 : 16);
}
// does a 3x3 launch of this shape run F(4x4,3x3)?  Decided from the shape alone: the weight images are packed with the
// same predicate (launch_pack / pack_all).  Default algorithm: maps of >= 60x80 pixels with >= 4 tile blocks per CU
// (measured 1.2x / 1.15x faster than F(2x2,3x3) on the 64 -> 64 layers at 240x320 / 120x160; the 60x80 layers of a
// 32-pair step gain 0.7 % of the step, the 30x40 ones nothing).
static bool w4_eligible(const ssp_handle* h, int nprob, int N, int H, int W, int cin, int cout) {
  if (g_conv_algo != 1 && g_conv_algo != 10 && g_conv_algo != 11) return false;
  if (cin % 8 != 0 || cin > W4_MAX_CIN) return false;   // (the producer's BatchNorm scale | shift of <= W4_MAX_CIN channels sit in LDS)
  if (g_conv_algo == 10) return true;
  bool wide; int ty, tx;
  w4_geometry(H, W, wide, ty, tx);
  const long items = (long)nprob * N * ty * tx * cdiv(cout, NB);
  static const long min_px = getenv("SSP_W4_MIN_PIXELS") ? atol(getenv("SSP_W4_MIN_PIXELS")) : 60L * 80L;  // (perf-debug override)
  static const long min_items_x4 = getenv("SSP_W4_MIN_ITEMS_X4") ? atol(getenv("SSP_W4_MIN_ITEMS_X4")) : 16L;  // (quarter blocks per CU)
  return (long)H * W >= min_px && 4L * items >= min_items_x4 * (h ? h->n_cu : 256);
}
static bool conv_uses_w4(const ssp_handle* h, const ConvCall& c) {
  if (!(c.wino && c.allow_w4 && c.ks == 3 && c.in_mode != 2)) return false;
  if (c.force_w4 >= 0) return c.force_w4 == 1;  // conv_wino4_kernel is legal on every map size (algorithm 10 runs it everywhere)
  return w4_eligible(h, c.nprob, c.N, c.H, c.W, c.cin, c.cout);
}
static bool conv_uses_p2(const ssp_handle* h, const ConvCall& c) {
  if (!c.wino) return false;
  if (g_conv_algo == 6) return true;
  if (!pipe_algo() || conv_uses_w4(h, c)) return false;
  const bool w1 = (c.W % 32) == 0;
  const long items = (long)c.nprob * c.N * cdiv(c.H, w1 ? 8 : 32) * cdiv(c.W, w1 ? 32 : 8) * c.ncob;
  return items < 4L * (h ? h->n_cu : 256);
}
// will this forward launch write ConvCall::pool_out (first-generation pipelined Winograd kernel, contiguous output)?
static bool conv_writes_pool(const ssp_handle* h, const ConvCall& c) {
  // (first-generation pipelined kernels: fp32 algorithms 1 / 5 and the bf16-operand kernels 3 / 7 / 8, which share the tile geometry)
  return c.wino && c.in_mode == 1 && (pipe_algo() || g_conv_algo == 5 || bf16_algo()) && !conv_uses_p2(h, c) && c.H % 2 == 0 && c.W % 2 == 0 &&
         c.cout % 4 == 0 && c.out_co == 0 && c.out_cs == c.cout && (!conv_uses_w4(h, c) || c.cout % NB == 0);
}

// will this launch run a kernel whose prologue can derive its input layer's BatchNorm affine from the raw statistics (BnLazy:
// conv_wino4 / conv_wino_p2 / conv_wino_pipe reading under BatchNorm + ReLU)?  SSP_BN_LAZY=0 (perf-debug): bn_finalize_kernel everywhere.
static bool bn_lazy_env() {
  static const int v = getenv("SSP_BN_LAZY") ? atoi(getenv("SSP_BN_LAZY")) : 1;
  return v != 0;
}
static bool conv_has_bn_lazy(const ssp_handle* h, const ConvCall& c) {
  if (!bn_lazy_env() || !c.wino || c.in_mode != 1 || bf16_algo() || g_conv_algo == 5) return false;
  return conv_uses_w4(h, c) || conv_uses_p2(h, c) || pipe_algo();
}

static int launch_conv(ssp_handle* h, const ConvCall& c, hipStream_t st, int prof_family = 0) {
  ConvArgs a;
  a.in = c.in; a.wpk = c.wpk; a.bias = c.bias; a.out = c.out; a.in_scale = c.in_scale; a.in_shift = c.in_shift;
  a.stats = c.stats; a.N = c.N; a.H = c.H; a.W = c.W; a.Cin = c.cin; a.in_cs = c.in_cs; a.in_co = c.in_co;
  a.Cout = c.cout; a.out_cs = c.out_cs; a.out_co = c.out_co; a.nchunks = c.nchunks; a.ncob = c.ncob;
  a.nprob = c.nprob; a.in2 = c.in2; a.out2 = c.out2; a.in_scale2 = c.in_scale2; a.in_shift2 = c.in_shift2;
  a.stats2 = c.stats2;
  if (c.lazy.mode != 0) {
    if (!conv_has_bn_lazy(h, c)) return fail(-3, "consumer-side BatchNorm finalize needs one of the Winograd kernels of the default algorithm");
    a.lazy = c.lazy;
  }
  if (c.bnr_mode != 0) {
    if (!can_fuse_bnr(c)) return fail(-3, "fused BatchNorm-backward sums need the pipelined Winograd kernel");
    a.bnr_mode = c.bnr_mode; a.bnr_t = c.bnr_t[0]; a.bnr_t2 = c.bnr_t[1]; a.bnr_cs = c.bnr_cs; a.bnr_co = c.bnr_co;
    for (int k = 0; k < 2; ++k) {
      a.bnr_p0[k] = c.bnr_p[0][k]; a.bnr_p1[k] = c.bnr_p[1][k]; a.bnr_p2[k] = c.bnr_p[2][k]; a.bnr_p3[k] = c.bnr_p[3][k];
    }
  }
  {
    // the input is read through one buffer descriptor PER IMAGE (32-bit byte offsets inside it); element indices of
    // whole tensors are 32-bit in the element-wise kernels
    const double in_img = (double)c.H * c.W * (c.in_mode == 2 ? 4 : 1) * c.in_cs * 4.0;
    const double in_el = (double)c.N * c.H * c.W * (c.in_mode == 2 ? 4 : 1) * c.in_cs;
    const double out_el = (double)c.N * c.H * c.W * c.out_cs;
    if (in_img > 2147483647.0 || in_el > 2147483647.0 || out_el > 2147483647.0)
      return fail(-3, "conv tensor [%d,%d,%d,%d] exceeds 2^31 elements: lower the batch", c.N, c.H, c.W,
                  std::max(c.in_cs, c.out_cs));
    a.in_bytes = (unsigned)in_img;
    a.out_bytes = (unsigned)(out_el * 4.0 > 4294967295.0 ? 4294967295.0 : out_el * 4.0);
    a.wpk_bytes = (unsigned)((double)c.ncob * c.nchunks * (c.wino ? WC : c.ks * c.ks) * CK * NB * 4.0);
  }
  a.ablate = g_dbg_ablate;
  // tile geometry: second-generation Winograd kernel (algo 6): 32 tiles = 8x16 / 16x8 pixels per 4-wave workgroup;
  // everything else: 64 tiles (Winograd) or 256 pixels (direct) = 8x32 / 32x8 per workgroup
  // default algorithm (1): maps with few first-generation work items per CU (the 30x40 layers: 640 items on 256 CUs =
  // 2.5 rounds) run on the finer-grained second-generation kernel (measured 10-15 % faster there, 1-4 % slower on the
  // large maps: tools/archive/conv_probe.py)
  const bool p2 = conv_uses_p2(h, c);
  const bool w4 = conv_uses_w4(h, c);
  if (c.pool_out[0] != nullptr) {
    if (!conv_writes_pool(h, c)) return fail(-3, "pooled raw output needs the first-generation pipelined Winograd kernel");
    a.pool_out[0] = c.pool_out[0]; a.pool_out[1] = c.pool_out[1]; a.pool_gamma = c.pool_gamma;
  }
  const bool wide = p2 ? (c.W % 16) == 0 : (c.W % 32) == 0;
  const int TH = p2 ? (wide ? 8 : 16) : (wide ? 8 : 32), TW = p2 ? (wide ? 16 : 8) : (wide ? 32 : 8);
  a.tiles_x = cdiv(c.W, TW); a.tiles_y = cdiv(c.H, TH);
  bool wide4 = false;
  if (w4) {
    w4_geometry(c.H, c.W, wide4, a.tiles_y, a.tiles_x);
    a.wpk_bytes = (unsigned)((double)c.ncob * (c.cin / 8) * W4_B_FLOATS * 4.0);
  }
  // persistent grid: 2 blocks per CU (LDS-limited residency; first-generation Winograd: 1), a multiple of 8 (one slot
  // set per XCD)
  const int n_cu = h ? h->n_cu : 256;
  if (c.wino && (c.ks != 3 || c.in_mode == 2 || c.cin % CK)) return fail(-3, "Winograd conv needs ks 3, Cin %% 16 == 0");
  int nblocks = std::max(8, ((c.wino && !p2 ? 1 : 2) * n_cu) / 8 * 8);
  if (g_dbg_grid > 0) nblocks = g_dbg_grid;  // perf-debug only (ssp_debug_conv_knobs)
  if ((nblocks / 8) < c.ncob) return fail(-3, "too many output-channel blocks (%d) for the persistent grid", c.ncob);
  const double flops = 2.0 * c.nprob * c.N * c.H * c.W * (double)c.cin * c.cout * c.ks * c.ks;
  // algorithmic bytes: input + output once, plus what the FUSED passes of the launch move (round 6): the layer-below tensor of the
  // BatchNorm-backward sums (data gradient), the quarter-size raw pooled copy (forward)
  const double bytes = 4.0 * c.nprob * c.N * c.H * c.W * ((double)c.cin * (c.in_mode == 2 ? 4 : 1) + c.cout +
                                                          (c.bnr_mode != 0 ? (double)c.cout : 0.0) + (c.pool_out[0] != nullptr ? 0.25 * c.cout : 0.0));
  int fam = prof_family;
  if (prof_family == SSP_PROF_CONV3X3_FWD && h && h->prof_family == SSP_PROF_CONV_BIG_FWD && c.H * c.W >= 240 * 320 &&
      c.cin == 64)
    fam = SSP_PROF_CONV_BIG_FWD;
  // multiplies executed on the matrix cores: 36 per 16 outputs x 9 taps (F(4x4,3x3)), 16 per 4 x 9 (F(2x2,3x3))
  const int pkern = w4 ? SSP_PROF_K_CONV_WINO4 : (c.wino && bf16_algo()) ? SSP_PROF_K_OTHER : p2 ? SSP_PROF_K_CONV_WINO_P2
                    : (c.wino && (pipe_algo() || g_conv_algo == 5)) ? SSP_PROF_K_CONV_WINO_PIPE : SSP_PROF_K_OTHER;
  ProfScope ps(h, fam, st, flops, bytes, flops * (w4 ? 0.25 : c.wino ? 16.0 / 36.0 : 1.0), pkern);
  if (w4) {  // Winograd F(4x4,3x3), one 8-wave workgroup per CU
    if (c.in_mode == 0) return wide4 ? launch_wino4_t<0, true>(a, nblocks, st) : launch_wino4_t<0, false>(a, nblocks, st);
    return wide4 ? launch_wino4_t<1, true>(a, nblocks, st) : launch_wino4_t<1, false>(a, nblocks, st);
  }
#if SSP_LEGACY_ALGOS
  if (c.wino && bf16_algo()) {
    if (bf16_parts(c.backward) == 1) {
      a.wpk_bytes /= 2;  // one bf16 part: half the bytes of the fp32 image
      if (c.in_mode == 0) return wide ? launch_wino_bf16_t<0, true>(a, nblocks, st) : launch_wino_bf16_t<0, false>(a, nblocks, st);
      return wide ? launch_wino_bf16_t<1, true>(a, nblocks, st) : launch_wino_bf16_t<1, false>(a, nblocks, st);
    }
    // hi + lo parts (the image has the size of the fp32 one)
    if (c.in_mode == 0) return wide ? launch_wino_bf16_t<0, true, 2>(a, nblocks, st) : launch_wino_bf16_t<0, false, 2>(a, nblocks, st);
    return wide ? launch_wino_bf16_t<1, true, 2>(a, nblocks, st) : launch_wino_bf16_t<1, false, 2>(a, nblocks, st);
  }
#endif
  if (p2) {  // second-generation pipelined Winograd: two independent 4-wave workgroups per CU
    if (c.in_mode == 0) return wide ? launch_wino_p2_t<0, true>(a, nblocks, st) : launch_wino_p2_t<0, false>(a, nblocks, st);
    return wide ? launch_wino_p2_t<1, true>(a, nblocks, st) : launch_wino_p2_t<1, false>(a, nblocks, st);
  }
  if (c.wino && pipe_algo()) {  // pipelined Winograd, weight fragments straight from L2 (default)
    if (c.in_mode == 0) return wide ? launch_wino_pipe_t<0, true, true>(a, nblocks, st) : launch_wino_pipe_t<0, false, true>(a, nblocks, st);
    return wide ? launch_wino_pipe_t<1, true, true>(a, nblocks, st) : launch_wino_pipe_t<1, false, true>(a, nblocks, st);
  }
#if SSP_LEGACY_ALGOS
  if (c.wino && g_conv_algo == 5) {  // the same pipeline with the weights staged through LDS
    if (c.in_mode == 0) return wide ? launch_wino_pipe_t<0, true>(a, nblocks, st) : launch_wino_pipe_t<0, false>(a, nblocks, st);
    return wide ? launch_wino_pipe_t<1, true>(a, nblocks, st) : launch_wino_pipe_t<1, false>(a, nblocks, st);
  }
  if (c.wino) {
    if (c.in_mode == 0) return wide ? launch_wino_t<0, true>(a, nblocks, st) : launch_wino_t<0, false>(a, nblocks, st);
    return wide ? launch_wino_t<1, true>(a, nblocks, st) : launch_wino_t<1, false>(a, nblocks, st);
  }
#else
  if (c.wino) return fail(-3, "conv algorithm %d is compiled out (build with -DSSP_LEGACY_ALGOS=1)", g_conv_algo);
#endif
#define CONV_CASE(KS_, M_)                                                          \
  if (c.ks == KS_ && c.in_mode == M_) {                                             \
    return wide ? launch_conv_t<KS_, M_, 1, 32>(a, nblocks, st) : launch_conv_t<KS_, M_, 4, 8>(a, nblocks, st); \
  }
  CONV_CASE(3, 0) CONV_CASE(3, 1) CONV_CASE(3, 2) CONV_CASE(1, 0) CONV_CASE(1, 1)
#undef CONV_CASE
  return fail(-3, "unsupported conv variant ks=%d in_mode=%d", c.ks, c.in_mode);
}

// ---- grouped pointwise convolutions (conv1x1_group.hip.h): several layers x views in one launch ----
struct G1Layer {
  const float* in[2] = {nullptr, nullptr}; int in_cs = 0, in_co = 0;
  float* out[2] = {nullptr, nullptr}; int out_cs = 0, out_co = 0;
  const float* wpk = nullptr;   // pack_g1_kernel's image of the whole layer
  const float* bias = nullptr;
  const float* scale[2] = {nullptr, nullptr}; const float* shift[2] = {nullptr, nullptr};  // BatchNorm + ReLU on load (in_mode 1)
  double* stats[2] = {nullptr, nullptr};                                                   // BatchNorm statistics of the output
  // data gradient: pass 1 of the BatchNorm backward of the layer below into stats[] (G1Prob::bnr_y); parameters [N]
  const float* bnr_y[2] = {nullptr, nullptr};
  const float* bnr_scale[2] = {nullptr, nullptr}; const float* bnr_shift[2] = {nullptr, nullptr};
  const float* bnr_mean[2] = {nullptr, nullptr}; const float* bnr_invstd[2] = {nullptr, nullptr};
  int K = 0, N = 0;
};
static inline size_t g1_image_floats(int K, int N) { return (size_t)cdiv(K, G1_KC) * cdiv(N, 32) * G1_TILE_FLOATS; }
// SSP_G1=0 (perf-debug) and the direct algorithm 0 keep the pointwise layers on conv_mfma_kernel<1, ...>
static bool g1_enabled() {
  static const int env = getenv("SSP_G1") ? atoi(getenv("SSP_G1")) : 1;
  return env != 0 && g_conv_algo != 0;
}
static inline int g1_ntmax(int K) { return std::min(G1_NT, 32 / cdiv(K, G1_KC)); }  // n-tiles whose weight image fits the 128 KB of LDS
static bool g1_fits(int K, int N, long npx, int in_cs, int out_cs, int nviews) {
  if (K < 1 || K > G1_KMAX || N < 1 || npx <= 0) return false;
  const long parts = cdiv(cdiv(N, 32), g1_ntmax(K));
  return parts * nviews <= G1_MAXP && (double)npx * in_cs * 4.0 < 2147483648.0 && (double)npx * out_cs * 4.0 < 2147483648.0;
}
static int launch_g1(const G1Layer* L, int nl, int nviews, long npx, int in_mode, int n_cu, hipStream_t st) {
  G1Args a;
  a.nprob = 0;
  long cost[G1_MAXP], total = 0;
  bool bnr = nl > 0 && L[0].bnr_y[0] != nullptr;
  for (int i = 0; i < nl; ++i) {
    const G1Layer& y = L[i];
    if ((y.bnr_y[0] != nullptr) != bnr || (bnr && in_mode != 0)) return fail(-1, "grouped pointwise launch: mixed BatchNorm-backward fusion");
    if (!g1_fits(y.K, y.N, npx, y.in_cs, y.out_cs, nviews)) return fail(-3, "pointwise conv %d -> %d does not fit the grouped kernel", y.K, y.N);
    const int ntt = cdiv(y.N, 32), nparts = cdiv(ntt, g1_ntmax(y.K));
    int t0 = 0;
    for (int part = 0; part < nparts; ++part) {
      const int size = ntt / nparts + (part < ntt % nparts ? 1 : 0);
      for (int k = 0; k < nviews; ++k) {
        if (a.nprob >= G1_MAXP) return fail(-3, "grouped pointwise launch: more than %d problems", G1_MAXP);
        G1Prob& q = a.p[a.nprob];
        q.in = y.in[k]; q.out = y.out[k]; q.wpk = y.wpk + (size_t)t0 * G1_TILE_FLOATS; q.bias = y.bias ? y.bias + 32 * t0 : nullptr;
        q.in_scale = y.scale[k]; q.in_shift = y.shift[k]; q.stats = y.stats[k] ? y.stats[k] + 32 * t0 : nullptr; q.stats_c = y.N;
        q.in_cs = y.in_cs; q.in_co = y.in_co; q.out_cs = y.out_cs; q.out_co = y.out_co + 32 * t0;
        q.K = y.K; q.N = std::min(y.N - 32 * t0, 32 * size); q.nchunks = cdiv(y.K, G1_KC); q.nt = size; q.nt_total = ntt;
        q.npx = (int)npx; q.wg0 = 0; q.nwg = 0;
        q.in_bytes = (unsigned)((size_t)npx * y.in_cs * 4); q.out_bytes = (unsigned)((size_t)npx * y.out_cs * 4);
        q.bnr_y = y.bnr_y[k];
        q.bnr_scale = bnr ? y.bnr_scale[k] + 32 * t0 : nullptr; q.bnr_shift = bnr ? y.bnr_shift[k] + 32 * t0 : nullptr;
        q.bnr_mean = bnr ? y.bnr_mean[k] + 32 * t0 : nullptr; q.bnr_invstd = bnr ? y.bnr_invstd[k] + 32 * t0 : nullptr;
        if (in_mode != 0 && (!q.in_scale || !q.in_shift)) return fail(-1, "pointwise conv: in_mode 1 needs scale / shift");
        // cycles of a wave per 32-pixel tile (every problem of a launch has the same pixels): 4 MFMAs of 64 cycles per k-quad and
        // n-tile, plus the epilogue of an n-tile (16 stores and the channel sums; with the BatchNorm-backward sums 16 loads and the
        // gates as well) - without that term the 65-channel data gradient (9 k-quads) got half the CUs it needed
        static const long e_plain = getenv("SSP_G1_ECOST") ? atol(getenv("SSP_G1_ECOST")) : 2000;      // (perf-debug knobs)
        static const long e_bnr = getenv("SSP_G1_ECOST_BNR") ? atol(getenv("SSP_G1_ECOST_BNR")) : 5000;
        cost[a.nprob] = (long)size * (256L * cdiv(y.K, 4) + (bnr ? e_bnr : e_plain));
        total += cost[a.nprob];
        ++a.nprob;
      }
      t0 += size;
    }
  }
  if (a.nprob == 0) return 0;
  // One 8-wave workgroup per CU; the CUs are shared out in proportion to the MFMA counts (largest remainders), at least one
  // workgroup per problem, no more workgroups than a problem has 8-tile groups.
  const int grid = std::max(n_cu, a.nprob);
  const int max_wg = std::max(1, cdiv(cdiv(npx, G1_PX), G1_WAVES));
  int given = 0;
  double frac[G1_MAXP];
  for (int i = 0; i < a.nprob; ++i) {
    const double share = (double)grid * cost[i] / total;
    a.p[i].nwg = std::max(1, std::min((int)share, max_wg));
    frac[i] = share - (int)share;
    given += a.p[i].nwg;
  }
  while (given < grid) {
    int best = -1;
    for (int i = 0; i < a.nprob; ++i)
      if (a.p[i].nwg < max_wg && (best < 0 || frac[i] > frac[best])) best = i;
    if (best < 0) break;
    a.p[best].nwg += 1; frac[best] -= 1.0; ++given;
  }
  while (given > grid) {
    int best = 0;
    for (int i = 1; i < a.nprob; ++i)
      if (a.p[i].nwg > a.p[best].nwg) best = i;
    if (a.p[best].nwg <= 1) break;
    a.p[best].nwg -= 1; --given;
  }
  int wg = 0;
  for (int i = 0; i < a.nprob; ++i) { a.p[i].wg0 = wg; wg += a.p[i].nwg; }
  static AttrOnce attr0, attr1, attr2;
#define SSP_G1_LAUNCH(M_, B_, ATTR_)                                                                                                  \
  {                                                                                                                                   \
    auto kern = conv1x1_group_kernel<M_, B_>;                                                                                         \
    if (ATTR_.need()) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G1_LDS_BYTES)); \
    hipLaunchKernelGGL(kern, dim3(wg), dim3(64 * G1_WAVES), G1_LDS_BYTES, st, a);                                                     \
  }
  if (in_mode != 0) SSP_G1_LAUNCH(1, false, attr1)
  else if (bnr) SSP_G1_LAUNCH(0, true, attr2)
  else SSP_G1_LAUNCH(0, false, attr0)
#undef SSP_G1_LAUNCH
  HIPCHK(hipGetLastError());
  return 0;
}
// grouped weight gradient of pointwise layers with 256 input channels (wgrad1x1_group_kernel + wgrad1x1_reduce_kernel)
struct G1WLayer {
  const float* x[2] = {nullptr, nullptr}; int x_cs = 0, x_co = 0;
  const float* scale[2] = {nullptr, nullptr}; const float* shift[2] = {nullptr, nullptr};
  const float* dy[2] = {nullptr, nullptr}; int dy_cs = 0, dy_co = 0;
  float* dw = nullptr;  // OIHW [N][256], accumulated
  int N = 0;
};
static bool g1w_fits(int cin, int cout, long npx, int x_cs, int dy_cs) {
  return cin == 256 && cout >= 1 && npx >= 2 && (double)npx * x_cs * 4.0 < 2147483648.0 && (double)npx * dy_cs * 4.0 < 2147483648.0;
}
static int launch_g1_wgrad(const G1WLayer* L, int nl, int nviews, long npx, float* partial, int partial_slabs, int n_cu, hipStream_t st,
                           TailJobs* defer = nullptr) {   // defer: the slab reduction joins the tail launch of the backward pass
  G1WArgs a;
  a.nparts = 0;
  a.partial = partial;
  long cost[G1W_MAXP], total = 0;
  for (int i = 0; i < nl; ++i) {
    const G1WLayer& y = L[i];
    const int ntt = cdiv(y.N, 32), nparts = cdiv(ntt, G1_NT);
    int t0 = 0;
    for (int part = 0; part < nparts; ++part) {
      if (a.nparts >= G1W_MAXP) return fail(-3, "grouped pointwise weight gradient: more than %d parts", G1W_MAXP);
      const int size = ntt / nparts + (part < ntt % nparts ? 1 : 0);  // balanced (133 outputs: 3 + 2 n-tiles; a 1-tile part
                                                                      // would issue one load per MFMA)
      G1WPart& q = a.p[a.nparts];
      for (int k = 0; k < 2; ++k) {
        const int kk = k < nviews ? k : 0;
        q.x[k] = y.x[kk]; q.scale[k] = y.scale[kk]; q.shift[k] = y.shift[kk]; q.dy[k] = y.dy[kk];
      }
      q.dw = y.dw + (size_t)32 * t0 * 256;
      q.x_cs = y.x_cs; q.x_co = y.x_co; q.dy_cs = y.dy_cs; q.dy_co = y.dy_co + 32 * t0;
      q.N = std::min(y.N - 32 * t0, 32 * size); q.nt = size; q.npx = (int)npx; q.nviews = nviews;
      cost[a.nparts] = size; total += size;
      ++a.nparts;
      t0 += size;
    }
  }
  if (a.nparts == 0) return 0;
  // grid shares proportional to the MFMA count of a part, a multiple of the view count each
  const int grid_max = std::min(2 * n_cu, partial_slabs);
  const long pairs = (npx + 1) / 2;
  int wg = 0;
  for (int i = 0; i < a.nparts; ++i) {
    long n = (long)grid_max * cost[i] / total / nviews * nviews;
    n = std::max<long>(nviews, std::min<long>(n, pairs * nviews));
    a.p[i].wg0 = wg; a.p[i].nwg = (int)n;
    wg += (int)n;
  }
  if (wg > partial_slabs) return fail(-4, "grouped pointwise weight gradient: %d partial slabs needed, %d available", wg, partial_slabs);
  hipLaunchKernelGGL(wgrad1x1_group_kernel, dim3(wg), dim3(256), 0, st, a);
  int rblocks = 0;
  for (int i = 0; i < a.nparts; ++i) rblocks += 32 * a.p[i].nt;
  if (defer != nullptr) { defer->g1 = a; defer->g1_blocks = rblocks; }
  else hipLaunchKernelGGL(wgrad1x1_reduce_kernel, dim3(rblocks), dim3(256), 0, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}
// one launch packs the operand images of up to G1_PACK_MAX_JOBS (layer, direction) pairs
static void g1_add_pack(G1PackJobs& J, int& nblocks, const float* w, float* dst, int cout_w, int cin_w, int transpose) {
  G1PackJob& q = J.j[J.n++];
  q.w = w; q.dst = dst; q.cout_w = cout_w; q.cin_w = cin_w; q.transpose = transpose;
  const int K = transpose ? cout_w : cin_w, N = transpose ? cin_w : cout_w;
  q.nchunks = cdiv(K, G1_KC); q.nt_total = cdiv(N, 32); q.block0 = nblocks;
  nblocks += cdiv((long)q.nchunks * q.nt_total * G1_TILE_FLOATS, 256);
}

template <int KS, int IN_MODE, int SH, int SW>
static int launch_wgrad_t(const WgradArgs& a, int nblocks, hipStream_t st) {
  using G = WgradGeom<KS, SH, SW>;
  static AttrOnce attr_once;
  auto kern = wgrad_mfma_kernel<KS, IN_MODE, SH, SW>;
  if (attr_once.need()) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), G::LDS_BYTES, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

template <int IN_MODE, bool WIDE>
static int launch_wgrad_wino_t(const WgradArgs& a, int nblocks, hipStream_t st) {
  using G = WgradWinoGeom<WIDE>;
  static AttrOnce attr_once;
  auto kern = wgrad_wino_kernel<IN_MODE, WIDE>;
  if (attr_once.need()) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(512), G::LDS_BYTES, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

template <int IN_MODE, bool WIDE, bool POOL, bool BF16 = false>
static int launch_wgrad_wino_fused_t(const WgradArgs& a_in, int nblocks, hipStream_t st) {
  using GF = WgradFusedGeom<WIDE>;
  static AttrOnce attr_once;
  auto kern = wgrad_wino_fused_kernel<IN_MODE, WIDE, POOL, BF16>;
  if (attr_once.need()) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, GF::LDS_BYTES));
  }
#if WGF_TRACE
  // perf-debug build only: every SSP_WGF_TRACE-th launch carries a trace buffer, is waited for and printed (cycles per tile)
  WgradArgs a = a_in;
  static const int trace_env = getenv("SSP_WGF_TRACE") ? atoi(getenv("SSP_WGF_TRACE")) : 0;
  static unsigned long long* trace_buf = nullptr;
  static int trace_count = 0;
  const bool trace_now = trace_env > 0 && (++trace_count % trace_env) == 0;
  if (trace_now) {
    if (!trace_buf) HIPCHK(hipMalloc(&trace_buf, 64 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(trace_buf, 0, 64 * sizeof(unsigned long long), st));
    a.trace = trace_buf;
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(512), GF::LDS_BYTES, st, a);
  HIPCHK(hipGetLastError());
  if (trace_now) {
    unsigned long long hbuf[64];
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(hbuf, trace_buf, sizeof(hbuf), hipMemcpyDeviceToHost));
    const double nt = (double)std::max<unsigned long long>(hbuf[6], 1);
    fprintf(stderr, "[wgf trace] %dx%d cin %d cout %d mode %d pool %d nprob %d: %llu tiles of workgroup 0 at %.0f MHz; cycles per tile (MFMA floor 2 x 8192 per SIMD)\n",
            a.H, a.W, a.Cin, a.Cout, IN_MODE, (int)POOL, a.nprob, hbuf[6], 100.0 * (double)hbuf[5] / (double)std::max<unsigned long long>(hbuf[7], 1));
    for (int w = 0; w < 8; ++w)
      fprintf(stderr, "  wave %d: top barrier %.0f  staging %.0f  barrier %.0f  issue %.0f  mfma %.0f  | loop %.0f\n", w, hbuf[w * 8] / nt,
              hbuf[w * 8 + 1] / nt, hbuf[w * 8 + 2] / nt, hbuf[w * 8 + 3] / nt, hbuf[w * 8 + 4] / nt, hbuf[w * 8 + 5] / nt);
  }
#else
  const WgradArgs& a = a_in;
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(512), GF::LDS_BYTES, st, a);
  HIPCHK(hipGetLastError());
#endif
  return 0;
}

template <int IN_MODE, bool WIDE>
static int launch_wgrad_wino4_t(const WgradArgs& a, int nblocks, hipStream_t st) {
  using G = Wgrad4Geom<WIDE>;
  static AttrOnce attr_once;
  auto kern = wgrad_wino4_kernel<IN_MODE, WIDE>;
  if (attr_once.need()) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(WG4_THREADS), G::LDS_BYTES, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}

#if SSP_LEGACY_ALGOS
template <int IN_MODE, bool WIDE, int NT = 1>
static int launch_wgrad_wino_bf16_t(const WgradArgs& a, int nblocks, hipStream_t st) {
  using G = WgradWinoGeom<WIDE>;
  static AttrOnce attr_once;
  auto kern = wgrad_wino_bf16_kernel<IN_MODE, WIDE, NT>;
  if (attr_once.need()) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(512), G::LDS_BYTES, st, a);
  HIPCHK(hipGetLastError());
  return 0;
}
#endif

struct WgradCall {
  const float* in; int in_cs, in_co, cin;
  const float* dout; int dout_cs, dout_co, cout;
  const float* in_scale; const float* in_shift;
  float* dw;  // OIHW gradient, accumulated
  int N, H, W, ks, in_mode;
  int nprob = 1;  // optional second problem accumulated into the same gradient
  const float* in2 = nullptr; const float* dout2 = nullptr;
  const float* in_scale2 = nullptr; const float* in_shift2 = nullptr;
  // fused BatchNorm + ReLU + MaxPool backward APPLY (wgrad_wino_fused_kernel, WgradArgs::f_*): dout / dout2 = pooled gradient
  bool fuse_apply = false, fuse_pool = false;  // fuse_pool: the layer is followed by MaxPool2d(2) (dout = pooled gradient)
  const float* f_y[2] = {nullptr, nullptr};
  float* f_dy[2] = {nullptr, nullptr};
  const float* f_scale[2] = {nullptr, nullptr}; const float* f_shift[2] = {nullptr, nullptr};
  const float* f_mean[2] = {nullptr, nullptr}; const float* f_invstd[2] = {nullptr, nullptr};
  const float* f_k12[2] = {nullptr, nullptr};
  const float* f_gamma = nullptr;
  int f_ycs = 0;
  int f_lazy = 0;   // WgradArgs::f_lazy and its arguments
  const double* f_bsums[2] = {nullptr, nullptr};
  double f_count = 0.0;
  float* f_dgamma = nullptr; float* f_dbeta = nullptr; float* f_dbias = nullptr;
};

// Can the weight gradient of a layer whose output feeds BatchNorm + ReLU + MaxPool2d(2) take that layer's APPLY pass along
// (wgrad_wino_fused_kernel)?  fp32 Winograd F(3x3,2x2) weight gradient, even map, whole 4-channel quads, dense y / dY.
static bool wgrad_can_fuse_apply(int ks, int in_mode, int H, int W, int cout) {
  static const int env = getenv("SSP_FUSE_APPLY") ? atoi(getenv("SSP_FUSE_APPLY")) : 1;  // (perf-debug A/B)
  static const int f4 = getenv("SSP_WGRAD_F4") ? atoi(getenv("SSP_WGRAD_F4")) : 0;        // (forces wgrad_wino4_kernel)
  const bool algo_ok = g_conv_algo == 1 || g_conv_algo == 9 || g_conv_algo == 10 || (bf16_algo() && bf16_parts(true) == 1);
  return env != 0 && f4 == 0 && algo_ok && ks == 3 && in_mode != 2 && H % 2 == 0 && W % 2 == 0 && cout % 4 == 0;
}

// sums the pending partial slabs of the deferred Winograd weight-gradient launches into the OIHW gradients
static int flush_wgrad_reduce_bf16(ssp_handle* h, hipStream_t st);
// SSP_TAIL_MERGE=0 (perf-debug A/B): the pointwise slab reduction and the column sums as launches of their own
static bool tail_merge_env() {
  static const int v = getenv("SSP_TAIL_MERGE") ? atoi(getenv("SSP_TAIL_MERGE")) : 1;
  return v != 0;
}
static int flush_wgrad_reduce(ssp_handle* h, hipStream_t st) {
  if (h != nullptr) CHK(flush_wgrad_reduce_bf16(h, st));
  if (h == nullptr) return 0;
  const int tail_blocks = h->tail.g1_blocks + h->tail.cs_blocks * h->tail.cs_views;
  if (h->rjobs.n == 0 && tail_blocks == 0) { h->partial_used = 0; return 0; }
  int nblocks = tail_blocks;
  if (h->rjobs.n != 0) {
    const WredJob& last = h->rjobs.j[h->rjobs.n - 1];
    nblocks += last.block0 + last.nblocks;
  }
  hipLaunchKernelGGL(wgrad_wino_reduce_multi_kernel, dim3(nblocks), dim3(256), 0, st, h->rjobs, h->tail);
  HIPCHK(hipGetLastError());
  h->rjobs.n = 0;
  h->tail.g1_blocks = 0; h->tail.cs_views = 0;
  h->partial_used = 0;
  return 0;
}

static int launch_wgrad(ssp_handle* h, const WgradCall& c, float* partial, size_t partial_floats, int n_cu,
                        hipStream_t st) {
  WgradArgs a;
  a.in = c.in; a.dout = c.dout; a.partial = partial; a.in_scale = c.in_scale; a.in_shift = c.in_shift;
  a.N = c.N; a.H = c.H; a.W = c.W; a.Cin = c.cin; a.in_cs = c.in_cs; a.in_co = c.in_co;
  a.Cout = c.cout; a.dout_cs = c.dout_cs; a.dout_co = c.dout_co;
  a.nprob = c.nprob; a.in2 = c.in2; a.dout2 = c.dout2; a.in_scale2 = c.in_scale2; a.in_shift2 = c.in_shift2;
  a.ablate = g_dbg_ablate;
  if (c.fuse_apply) {
    for (int k = 0; k < 2; ++k) {
      a.f_y[k] = c.f_y[k]; a.f_dy[k] = c.f_dy[k]; a.f_scale[k] = c.f_scale[k]; a.f_shift[k] = c.f_shift[k];
      a.f_mean[k] = c.f_mean[k]; a.f_invstd[k] = c.f_invstd[k]; a.f_k12[k] = c.f_k12[k];
    }
    a.f_gamma = c.f_gamma; a.f_ycs = c.f_ycs;
    a.f_lazy = c.f_lazy; a.f_bsums[0] = c.f_bsums[0]; a.f_bsums[1] = c.f_bsums[1]; a.f_count = c.f_count;
    a.f_dgamma = c.f_dgamma; a.f_dbeta = c.f_dbeta; a.f_dbias = c.f_dbias;
  }
  const bool wide = (c.W % 32) == 0;
  // Winograd F(3x3,2x2): 3x3 filters on even-sized maps with a prefetchable (non-pooled) input
  // Winograd F(3x3,4x4) (wgrad_wino4_kernel: 1/4 of the direct multiplies), OPT-IN: conv algorithm 11 (= algorithm 1 with this
  // weight gradient) or SSP_WGRAD_F4=1 under algorithms 1 / 10.  Correct on every 3x3 layer with a prefetchable input and any
  // map size (dY is zero-filled outside the map); 64 ci x 32 co slabs over the same 128-pixel block tiles as F(3x3,2x2).
  // Not the default: measured equal at 240x320 and 10-50 % slower below (profiles/r03_kernel_experiments.txt, DESIGN.md s. 12).
  static const int wg4_env = getenv("SSP_WGRAD_F4") ? atoi(getenv("SSP_WGRAD_F4")) : 0;
  const bool wino4 = (g_conv_algo == 11 || (wg4_env != 0 && (g_conv_algo == 1 || g_conv_algo == 10))) && c.ks == 3 && c.in_mode != 2;
  const bool wino = wino4 || (g_conv_algo != 0 && c.ks == 3 && c.in_mode != 2 && c.H % 2 == 0 && c.W % 2 == 0);
  if (wino && ((double)c.H * c.W * std::max(c.in_cs, c.dout_cs) * 4.0 > 2147483647.0))
    return fail(-3, "wgrad: one image [%d,%d,%d] exceeds 2 GiB", c.H, c.W, std::max(c.in_cs, c.dout_cs));
  const int TH = wino ? (wide ? 4 : 16) : (wide ? 2 : 8), TW = wide ? 32 : 8;
  a.tiles_x = cdiv(c.W, TW); a.tiles_y = cdiv(c.H, TH);
  a.ntiles = c.N * a.tiles_x * a.tiles_y;
  a.ncib = cdiv(c.cin, 64); a.ncob = cdiv(c.cout, wino4 ? 32 : 64);
  const int pairs = a.ncib * a.ncob;
  const int taps = wino ? WC : c.ks * c.ks;  // 64 x 64 slabs per partial block
  const bool fused12 = c.fuse_apply && wino && !wino4;  // wgrad_wino_fused_kernel applies the right-hand product of G^T M G itself: 12-component slabs
  const size_t slab = wino4 ? (size_t)WG4_SLAB : fused12 ? (size_t)12 * 4096 : (size_t)taps * 4096;  // floats per partial block
  int nsplit = ((wino ? 1 : 2) * n_cu) / pairs / 8 * 8;  // blocks per CU; multiple of 8: blocks sharing tiles share an XCD
  if (nsplit < 1) nsplit = 1;
  if (nsplit > a.ntiles * a.nprob) nsplit = a.ntiles * a.nprob;
  while ((size_t)pairs * nsplit * slab > partial_floats && nsplit > 1) --nsplit;
  if ((size_t)pairs * nsplit * slab > partial_floats) return fail(-4, "wgrad scratch too small");
  a.nsplit = nsplit;
  const int nblocks = pairs * nsplit;
  // engine launches (partial == the handle's buffer): Winograd slabs stay in their own slice until flush_wgrad_reduce
  const bool deferred = wino && h != nullptr && partial == h->partial;
  if (deferred) {
    const size_t need = (size_t)pairs * nsplit * slab;
    if (h->partial_used + need > partial_floats || h->rjobs.n == WRED_MAX_JOBS) CHK(flush_wgrad_reduce(h, st));
    a.partial = partial + h->partial_used;
    WredJob& q = h->rjobs.j[h->rjobs.n];
    q.partial = a.partial; q.dw = c.dw; q.cin = c.cin; q.cout = c.cout; q.ncob = a.ncob; q.nsplit = nsplit; q.f4 = wino4 ? 1 : fused12 ? 2 : 0;
    q.nblocks = a.ncob * c.cin * (fused12 ? 3 : 1);
    q.block0 = h->rjobs.n ? h->rjobs.j[h->rjobs.n - 1].block0 + h->rjobs.j[h->rjobs.n - 1].nblocks : 0;
    ++h->rjobs.n;
    h->partial_used += need;
  } else if (h != nullptr && partial == h->partial && h->rjobs.n > 0) {
    CHK(flush_wgrad_reduce(h, st));  // an immediate-reduce launch reuses the buffer from its start
  }
  {
    const double flops = 2.0 * c.nprob * c.N * c.H * c.W * (double)c.cin * c.cout * c.ks * c.ks;
    // algorithmic bytes: X + dY once; a launch that carries the layer's BatchNorm-backward APPLY reads y and the gradient wrt the
    // (pooled) activation INSTEAD of dY and writes dY for the data gradient (round 6: counted)
    const double bytes = 4.0 * c.nprob * c.N * c.H * c.W * ((double)c.cin * (c.in_mode == 2 ? 4 : 1) +
                                                            (c.fuse_apply ? (2.0 + (c.fuse_pool ? 0.25 : 1.0)) * c.cout : (double)c.cout));
    ProfScope ps(h, c.ks == 3 ? SSP_PROF_CONV3X3_WGRAD : -1, st, flops, bytes, flops * (wino4 ? 0.25 : wino ? 16.0 / 36.0 : 1.0),
                 wino4 ? SSP_PROF_K_WGRAD_WINO4 : wino && !bf16_algo() ? SSP_PROF_K_WGRAD_WINO : SSP_PROF_K_OTHER);
    if (c.fuse_apply) {
      const bool b16 = bf16_algo();
      if (!wino || wino4 || (b16 && bf16_parts(true) != 1))
        return fail(-3, "fused BatchNorm apply needs the F(3x3,2x2) weight gradient (fp32, or bf16 operands in one part)");
#if SSP_LEGACY_ALGOS
#define WGF_CASE(M_, P_) \
      if (c.in_mode == M_ && c.fuse_pool == P_ && !b16) CHK((wide ? launch_wgrad_wino_fused_t<M_, true, P_>(a, nblocks, st) : launch_wgrad_wino_fused_t<M_, false, P_>(a, nblocks, st))); \
      if (c.in_mode == M_ && c.fuse_pool == P_ && b16) CHK((wide ? launch_wgrad_wino_fused_t<M_, true, P_, true>(a, nblocks, st) : launch_wgrad_wino_fused_t<M_, false, P_, true>(a, nblocks, st)));
#else
#define WGF_CASE(M_, P_) \
      if (c.in_mode == M_ && c.fuse_pool == P_) CHK((wide ? launch_wgrad_wino_fused_t<M_, true, P_>(a, nblocks, st) : launch_wgrad_wino_fused_t<M_, false, P_>(a, nblocks, st)));
#endif
      WGF_CASE(0, false) WGF_CASE(0, true) WGF_CASE(1, false) WGF_CASE(1, true)
#undef WGF_CASE
    } else if (wino4) {
      if (c.in_mode == 0) CHK((wide ? launch_wgrad_wino4_t<0, true>(a, nblocks, st) : launch_wgrad_wino4_t<0, false>(a, nblocks, st)));
      else CHK((wide ? launch_wgrad_wino4_t<1, true>(a, nblocks, st) : launch_wgrad_wino4_t<1, false>(a, nblocks, st)));
#if SSP_LEGACY_ALGOS
    } else if (wino && bf16_algo() && bf16_parts(true) == 1) {
      if (c.in_mode == 0) CHK((wide ? launch_wgrad_wino_bf16_t<0, true>(a, nblocks, st) : launch_wgrad_wino_bf16_t<0, false>(a, nblocks, st)));
      else CHK((wide ? launch_wgrad_wino_bf16_t<1, true>(a, nblocks, st) : launch_wgrad_wino_bf16_t<1, false>(a, nblocks, st)));
    } else if (wino && bf16_algo()) {
      if (c.in_mode == 0) CHK((wide ? launch_wgrad_wino_bf16_t<0, true, 2>(a, nblocks, st) : launch_wgrad_wino_bf16_t<0, false, 2>(a, nblocks, st)));
      else CHK((wide ? launch_wgrad_wino_bf16_t<1, true, 2>(a, nblocks, st) : launch_wgrad_wino_bf16_t<1, false, 2>(a, nblocks, st)));
#endif
    } else if (wino) {
      if (c.in_mode == 0) CHK((wide ? launch_wgrad_wino_t<0, true>(a, nblocks, st) : launch_wgrad_wino_t<0, false>(a, nblocks, st)));
      else CHK((wide ? launch_wgrad_wino_t<1, true>(a, nblocks, st) : launch_wgrad_wino_t<1, false>(a, nblocks, st)));
    }
    if (wino) {
      if (!deferred && wino4)
        hipLaunchKernelGGL(wgrad_wino4_reduce_kernel, dim3(a.ncob * c.cin), dim3(256), 0, st, partial, c.dw, c.cin, c.cout,
                           a.ncob, nsplit);
      else if (!deferred && fused12)
        hipLaunchKernelGGL(wgrad_fused12_reduce_kernel, dim3(3 * a.ncob * c.cin), dim3(256), 0, st, partial, c.dw, c.cin, c.cout,
                           a.ncob, nsplit);
      else if (!deferred)
        hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3(a.ncob * c.cin), dim3(256), 0, st, partial, c.dw, c.cin, c.cout,
                           a.ncob, nsplit);
      HIPCHK(hipGetLastError());
      return 0;
    }
#define WG_CASE(KS_, M_)                                                              \
  if (c.ks == KS_ && c.in_mode == M_) {                                               \
    CHK((wide ? launch_wgrad_t<KS_, M_, 1, 32>(a, nblocks, st) : launch_wgrad_t<KS_, M_, 4, 8>(a, nblocks, st))); \
  } else
    WG_CASE(3, 0) WG_CASE(3, 1) WG_CASE(3, 2) WG_CASE(1, 0) WG_CASE(1, 1)
    return fail(-3, "unsupported wgrad variant ks=%d in_mode=%d", c.ks, c.in_mode);
#undef WG_CASE
  }
  const int total = c.cout * c.cin * taps;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 64)), dim3(256), 0, st, partial, c.dw, c.cin, c.cout, c.ks,
                     a.ncob, nsplit);
  HIPCHK(hipGetLastError());
  return 0;
}

static int launch_pack(const float* w, float* dst, int cout_w, int cin_w, int ks, int tf, bool wino, hipStream_t st, bool w4 = false) {
  const int taps = ks * ks;
  const int conv_cin = tf ? cout_w : cin_w, conv_cout = tf ? cin_w : cout_w;
  const int nchunks = cdiv(conv_cin, CK), ncob = cdiv(conv_cout, NB);
  if (wino) {
    const int total = ncob * nchunks * WB_FLOATS;
    if (w4)  // F(4x4,3x3) image of conv_wino4_kernel: 8-channel chunks of 36 components
      hipLaunchKernelGGL(pack_weights_wino4_kernel, dim3(cdiv(ncob * 2 * nchunks * W4_B_FLOATS, 256)), dim3(256), 0, st, w, dst,
                         cout_w, cin_w, tf, 2 * nchunks, 0, 0, ncob, 2 * nchunks);
#if SSP_LEGACY_ALGOS
    else if (bf16_algo())
      hipLaunchKernelGGL(pack_weights_wino8_bf16_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w,
                         reinterpret_cast<__bf16*>(dst), cout_w, cin_w, tf, 2 * nchunks, 0, 0, ncob, 2 * nchunks,
                         bf16_parts(tf != 0));
#endif
    else if (pipe_algo() || g_conv_algo == 5 || g_conv_algo == 6)  // 8-channel stages of the pipelined kernels: twice as many chunks of half the size
      hipLaunchKernelGGL(pack_weights_wino8_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, dst, cout_w, cin_w, tf,
                         2 * nchunks, 0, 0, ncob, 2 * nchunks);
    else
      hipLaunchKernelGGL(pack_weights_wino_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, dst, cout_w, cin_w, tf,
                         nchunks, 0, 0, ncob, nchunks);
    HIPCHK(hipGetLastError());
    return 0;
  }
  const int total = ncob * nchunks * taps * CK * NB;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, dst, cout_w, cin_w, ks, tf,
                     nchunks, 0, 0, ncob, nchunks);
  HIPCHK(hipGetLastError());
  return 0;
}

struct BnPlainJob {   // arguments of one plain (no ReLU) BatchNorm backward, collected instead of launched (launch_bn_bwd_pair)
  BnBwdArgs a[2];
  float* dg = nullptr;
  float* db = nullptr;
  bool filled = false;
};
// Replica reductions of several layers whose pass 1 is complete at the same point of the backward pass (the 3x3 heads: pass 1 in the
// pointwise heads' data gradient, pass 2 in their weight gradients) collected for ONE launch - each small launch is ~13 us of the step.
struct BnSumsQueue {
  BnSumsJobs J;
  int maxC = 0;
  BnSumsQueue() { J.n = 0; }
};
static int flush_bn_sums(BnSumsQueue& q, int nviews, hipStream_t st) {
  if (q.J.n == 0) return 0;
  if (q.J.n == 1) hipLaunchKernelGGL(bn_bwd_sums_kernel<float>, dim3(cdiv(q.maxC * 32, 256)), dim3(256), 0, st, q.J.a0[0], q.J.a1[0], nviews,
                                     q.J.dgamma[0], q.J.dbeta[0]);
  else hipLaunchKernelGGL(bn_bwd_sums_multi_kernel<float>, dim3(cdiv(q.maxC * 32, 256), q.J.n), dim3(256), 0, st, q.J, nviews);
  HIPCHK(hipGetLastError());
  q.J.n = 0; q.maxC = 0;
  return 0;
}
// pass 1 (sums) -> replica reduction + dgamma/dbeta -> pass 2 (apply); a[0 .. nviews-1] ride the same launches
template <bool RELU, bool POOL, typename T = float>
static int launch_bn_bwd(const BnBwdArgs* a, int nviews, float* dgamma, float* dbeta, hipStream_t st,
                         bool sums_done = false, bool skip_apply = false, BnSumsQueue* queue = nullptr) {
  // (the reduction reads the tensors - as T - only for the pool_fix scan: without it the fp32 instantiation serves the bf16 path too)
  if (queue != nullptr && sums_done && skip_apply && (sizeof(T) == 4 || a[0].pool_fix == 0) && queue->J.n < 3) {  // only the reduction is left: the caller flushes
    const int j = queue->J.n++;
    queue->J.a0[j] = a[0]; queue->J.a1[j] = a[nviews - 1]; queue->J.dgamma[j] = dgamma; queue->J.dbeta[j] = dbeta;
    queue->maxC = std::max(queue->maxC, a[0].C);
    return 0;
  }
  // a.dbias (conv bias gradient, may be null) is produced by bn_bwd_sums_kernel
  const BnBwdArgs& a0 = a[0];
  const BnBwdArgs& a1 = a[nviews - 1];
  const int nq = (a0.C + 3) / 4, rows = 256 / nq;
  const long npix = (long)a0.N * (POOL ? a0.H / 2 : a0.H) * (POOL ? a0.W / 2 : a0.W);
  int nb = cdiv(npix, rows);
  if (nb > 1024) nb = 1024;
  if (!sums_done)  // else: pass 1 was accumulated by the data-gradient conv that produced dOut
    hipLaunchKernelGGL((bn_bwd_kernel<RELU, POOL, false, T>), dim3(nb, nviews), dim3(256), 0, st, a0, a1, a0, a1);
  hipLaunchKernelGGL(bn_bwd_sums_kernel<T>, dim3(cdiv(a0.C * 32, 256)), dim3(256), 0, st, a0, a1, nviews, dgamma, dbeta);
  if (!skip_apply)  // else: pass 2 rides the layer's weight gradient (wgrad_wino_fused_kernel)
    hipLaunchKernelGGL((bn_bwd_kernel<RELU, POOL, true, T>), dim3(nb, nviews), dim3(256), 0, st, a0, a1, a0, a1);
  HIPCHK(hipGetLastError());
  return 0;
}

// Grid of the row-based first-layer kernels: exactly the blocks that are resident at once (occupancy x CUs, split over
// the views riding blockIdx.y), the rows dealt round-robin.  The former fixed ~1000-block grid ran 1.25 / 2.5 / 3.75
// rounds of resident blocks: the partly filled last round cost conv0_direct_kernel a quarter of its time.
template <typename K>
static int l0_resident_grid(K kern, const ssp_handle* h, int nviews, long rows, int W) {
  static int per_cu = 0, per_cu_w = 0;  // per kernel (template instance) and image width: the query is not free
  if (per_cu_w != W) { per_cu = 0; per_cu_w = W; }
  if (per_cu == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, l0_lds_bytes(W)) != hipSuccess || per_cu < 1)) {
    (void)hipGetLastError();
    per_cu = -1;
  }
  if (per_cu < 1) return l0_grid(rows);
  const long g = std::max(1L, (long)per_cu * (h ? h->n_cu : 256) / std::max(1, nviews));
  return (int)std::min(rows, g);
}

#include "bf16_host.hip.h"

static int flush_wgrad_reduce_bf16(ssp_handle* h, hipStream_t st) { return h->rq_bf16 != nullptr ? h->rq_bf16->flush(st) : 0; }

static int ensure_aux_stream(ssp_handle* h);

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
// ---- bit-reproducible accumulation (csrc/det.hip.h): process-wide switch, fp32 scatter targets with fixed-point shadows ----
struct DetHostRegion { float* lo; size_t n; long long* shadow; const ssp_handle* owner; };
static std::vector<DetHostRegion> g_det_regions;
static int g_det_mode = -1;   // -1: not read yet (SSP_DETERMINISTIC), 0 / 1
static int g_det_device = -1; // device of the registered regions
static bool det_mode() {
  if (g_det_mode < 0) { const char* e = getenv("SSP_DETERMINISTIC"); g_det_mode = (e != nullptr && atoi(e) != 0) ? 1 : 0; }
  return g_det_mode == 1;
}
static int det_upload() {
  DetRegion tab[DET_MAX_REGIONS];
  int n = 0;
  for (const DetHostRegion& r : g_det_regions) {
    if (n == DET_MAX_REGIONS) return fail(-3, "deterministic mode: more than %d fp32 scatter targets are bound in this process", DET_MAX_REGIONS);
    tab[n].lo = r.lo; tab[n].hi = r.lo + r.n; tab[n].shadow = r.shadow; ++n;
  }
  for (int i = n; i < DET_MAX_REGIONS; ++i) { tab[i].lo = tab[i].hi = nullptr; tab[i].shadow = nullptr; }
  const int flag = det_mode() ? 1 : 0;
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_det_region), tab, sizeof(tab)));
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_det_nregion), &n, sizeof(n)));
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_det), &flag, sizeof(flag)));
  return 0;
}
static void det_release(const ssp_handle* h) {
  bool any = false;
  for (size_t i = 0; i < g_det_regions.size();) {
    if (g_det_regions[i].owner == h) { (void)hipFree(g_det_regions[i].shadow); g_det_regions.erase(g_det_regions.begin() + i); any = true; }
    else ++i;
  }
  // (an upload that fails here leaves the device table pointing at freed shadows only for regions nobody scatters into any more:
  // their owner is being re-bound or destroyed; the next successful upload replaces the table)
  if (any && det_upload() != 0) fprintf(stderr, "libssp_hip: deterministic-mode region table upload failed on release: %s\n", g_err.c_str());
}
static int det_register(const ssp_handle* h, float* lo, size_t n) {
  if (lo == nullptr || n == 0) return 0;
  DetHostRegion r;
  r.lo = lo; r.n = n; r.owner = h; r.shadow = nullptr;
  HIPCHK(hipMalloc(&r.shadow, n * sizeof(long long)));
  HIPCHK(hipMemset(r.shadow, 0, n * sizeof(long long)));
  g_det_regions.push_back(r);
  return 0;
}
// shadow of the region that starts at `base` -> the tensor (dst += shadow 2^-40; shadow = 0); a no-op outside deterministic mode
static int det_fold(float* base, hipStream_t st) {
  if (!det_mode() || base == nullptr) return 0;
  for (const DetHostRegion& r : g_det_regions)
    if (r.lo == base) {
      hipLaunchKernelGGL(det_fold_kernel, dim3((unsigned)std::min<long>(cdiv((long)r.n, 256), 2048)), dim3(256), 0, st, r.shadow, r.lo, (long)r.n);
      HIPCHK(hipGetLastError());
      return 0;
    }
  return 0;
}

extern "C" {

const char* ssp_last_error(void) { return g_err.c_str(); }

// sha256 of the library's sources at build time (hipbuild.source_id(): -DSSP_BUILD_ID); the marker prefix lets tools read it
// from the file without loading it
#ifndef SSP_BUILD_ID
#define SSP_BUILD_ID "unknown"
#endif
const char* ssp_build_id(void) {
  static const char marker[] = "SSP_BUILD_ID=" SSP_BUILD_ID;
  return marker + 13;
}

// Shader clock under matrix-core load (round 6: the benchmark line reports the clock its box sustains - boxes of the pool differ by
// +-3 % in pairs/s, mostly through the clock the power controller grants).  One workgroup of 4 waves per CU runs back-to-back
// v_mfma_f32_32x32x2_f32 on pseudo-random operands for ~`ms` milliseconds; workgroup 0 counts shader cycles (s_memtime) against the
// constant 100 MHz counter (s_memrealtime).
__global__ __launch_bounds__(256) void clock_probe_kernel(float* sink, unsigned long long* out, int iters) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[4], b[4];
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s = s * 1664525u + 1013904223u; a[i] = ((int)(s >> 8) - (1 << 23)) * (1.f / (1 << 23));
    s = s * 1664525u + 1013904223u; b[i] = ((int)(s >> 8) - (1 << 23)) * (0.05f / (1 << 23));
  }
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[(m + u) & 3], acc[m], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][9];
  sink[blockIdx.x * 256 + threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

int ssp_clock_probe(float ms, double* mhz_out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!mhz_out || !(ms > 0.f) || ms > 1000.f) return fail(-1, "ssp_clock_probe: bad argument");
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return fail(-2, "ssp_clock_probe: no device");
  float* sink = nullptr;
  unsigned long long* out = nullptr;
  if (hipMalloc(&sink, (size_t)ncu * 256 * 4) != hipSuccess) return fail(-2, "ssp_clock_probe: hipMalloc");
  if (hipMalloc(&out, 16) != hipSuccess) { (void)hipFree(sink); return fail(-2, "ssp_clock_probe: hipMalloc"); }
  // 16 MFMAs of 64 cycles per iteration and wave, one wave per SIMD: 1024 cycles per iteration; 2.4 GHz upper bound for the count
  const int iters = (int)(ms * 2.4e6f / 1024.f) + 1;
  unsigned long long h[2] = {0, 0};
  hipLaunchKernelGGL(clock_probe_kernel, dim3(ncu), dim3(256), 0, stream, sink, out, iters);
  hipError_t e = hipMemcpyAsync(h, out, 16, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(sink);
  (void)hipFree(out);
  if (e != hipSuccess || h[1] == 0) return fail(-2, "ssp_clock_probe: %s", hipGetErrorString(e));
  *mhz_out = 100.0 * (double)h[0] / (double)h[1];
  return 0;
}

// perf-debug (SSP_DBG_IDLE_US=n): one wave that sleeps for n microseconds of the constant 100 MHz counter - an idle stretch in front of
// the loss phase of the pair step, to see how much of it the power / clock management gives back to the kernels around it
// (PERF_LOG round 6 section 9).
__global__ void idle_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
}

int ssp_set_deterministic(int on) {
  g_det_mode = on ? 1 : 0;
  return det_upload();   // (handles bound before the switch keep their plain fp32 scatters until they are bound again)
}
int ssp_get_deterministic(void) { return det_mode() ? 1 : 0; }

int ssp_create(const ssp_config* cfg, ssp_handle** out) {
  if (!cfg || !out) return fail(-1, "null argument");
  if (cfg->arch != SSP_ARCH_GAUSS2 && cfg->arch != SSP_ARCH_GAUSS2_SSMALL) return fail(-1, "unknown arch %d", cfg->arch);
  if (cfg->height % 8 || cfg->width % 8 || cfg->height <= 0 || cfg->width <= 0)
    return fail(-1, "height/width must be positive multiples of 8 (got %dx%d)", cfg->height, cfg->width);
  if (cfg->width > L0_MAX_W) return fail(-1, "width must be <= %d (row images of the first-layer kernels)", L0_MAX_W);
  if (cfg->max_batch < 1 || cfg->max_batch > 1024) return fail(-1, "max_batch must be in 1..1024");
  if (cfg->arch == SSP_ARCH_GAUSS2_SSMALL && cfg->n_classes > 192) return fail(-1, "n_classes must be <= 192");
  ssp_handle* h = new ssp_handle();
  h->cfg = *cfg;
  if (h->cfg.n_classes <= 0) h->cfg.n_classes = 133;
  if (h->cfg.n_match <= 0) h->cfg.n_match = 1000;
  if (h->cfg.n_non <= 0) h->cfg.n_non = 100;
  h->conv_algo = g_default_conv_algo;
  h->rq_bf16 = new WredBQueue();
  AlgoScope algo(h);
  build_layers(h);
  h->bound = false;
  h->n_cu = device_cu_count();   // (before carve: the workspace layout depends on it)
  h->ws_bytes = carve(h, nullptr);
  h->prof_family = 0; h->ev_used = 0; h->prof_flops = h->prof_bytes = h->prof_exec_flops = 0; h->prof_launches = 0;
  *out = h;
  return 0;
}

void ssp_destroy(ssp_handle* h) {
  if (!h) return;
  det_release(h);
  delete h->rq_bf16;
  for (auto e : h->ev_pool) (void)hipEventDestroy(e);
  for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.exec);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->ev_pack_fork) (void)hipEventDestroy(h->ev_pack_fork);
  if (h->ev_pack_join) (void)hipEventDestroy(h->ev_pack_join);
  if (h->ev_early_fork) (void)hipEventDestroy(h->ev_early_fork);
  if (h->ev_early_join) (void)hipEventDestroy(h->ev_early_join);
  if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
  if (h->aux2_stream) (void)hipStreamDestroy(h->aux2_stream);
  delete h;
}

size_t ssp_param_count(const ssp_handle* h) { return h ? h->n_params : 0; }
size_t ssp_bn_channel_count(const ssp_handle* h) { return h ? h->n_bn_ch : 0; }
int ssp_bn_layer_count(const ssp_handle* h) { return h ? h->n_bn : 0; }
size_t ssp_workspace_bytes(const ssp_handle* h) { return h ? h->ws_bytes : 0; }

int ssp_bind(ssp_handle* h, const ssp_buffers* b, void* stream) {
  if (!h || !b) return fail(-1, "null argument");
  if (!b->params_dev || !b->bn_running_dev || !b->workspace_dev) return fail(-1, "params, bn_running and workspace are required");
  if (b->workspace_bytes < h->ws_bytes) return fail(-1, "workspace too small: %zu < %zu", b->workspace_bytes, h->ws_bytes);
  h->buf = *b;
  for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.exec);  // captured pointers are stale after a re-bind
  h->graphs.clear();
  carve(h, b->workspace_dev);
  CHK(ensure_aux_stream(h));  // (never created lazily inside a stream capture)
  // padding channels (65->80, n_classes->sout_cs) must read as zero forever: clear everything once
  HIPCHK(hipMemsetAsync(b->workspace_dev, 0, h->ws_bytes, (hipStream_t)stream));
  det_release(h);
  if (det_mode()) {   // fixed-point shadows of the tensors that fp32 atomics scatter into
    // The region table is ONE __device__ array per device while the host list is per process, and neither is locked: the mode
    // supports handles of one device, bound from one thread.  Capacity is checked BEFORE anything is registered, and a failure
    // half-way rolls this handle's regions back (a stale entry would fail every later bind of the process).
    int dev_now = -1;
    HIPCHK(hipGetDevice(&dev_now));
    if (g_det_device >= 0 && g_det_device != dev_now && !g_det_regions.empty())
      return fail(-3, "deterministic mode: handles of ONE device per process (regions of device %d are bound, this bind is on device %d)", g_det_device, dev_now);
    const int need = (b->grads_dev ? 1 : 0) + 2 + (h->slot[0].dsout ? 2 : 0);
    if ((int)g_det_regions.size() + need > DET_MAX_REGIONS)
      return fail(-3, "deterministic mode: %d fp32 scatter targets are bound in this process, this handle needs %d more (limit %d)",
                  (int)g_det_regions.size(), need, DET_MAX_REGIONS);
    g_det_device = dev_now;
    const size_t cells = (size_t)h->cfg.max_batch * (h->cfg.height / 8) * (h->cfg.width / 8);
    int rc = det_register(h, b->grads_dev, b->grads_dev ? h->n_params + 3 : 0);
    for (int v = 0; v < 2 && rc == 0; ++v) {
      rc = det_register(h, h->slot[v].ddesc, cells * 256);
      if (rc == 0) rc = det_register(h, h->slot[v].dsout, h->slot[v].dsout ? cells * h->sout_cs : 0);
    }
    if (rc == 0) rc = det_upload();
    if (rc != 0) { det_release(h); return rc; }
  }
  h->bound = true;
  return 0;
}

int ssp_zero_grad(ssp_handle* h, void* stream) {
  if (!h || !h->bound || !h->buf.grads_dev) return fail(-1, "handle not bound with a gradient buffer");
  CHK(dev_zero(h->buf.grads_dev, (h->n_params + 3) * sizeof(float), (hipStream_t)stream));
  return 0;
}

int ssp_profile_enable(ssp_handle* h, int family) {
  if (!h) return fail(-1, "null handle");
  h->prof_family = family; h->ev_used = 0; h->prof_flops = h->prof_bytes = h->prof_exec_flops = 0; h->prof_launches = 0;
  h->prof_paused = false;
  for (auto& k : h->prof_k) k = ssp_handle::ProfKernel();
  if (family != 0 && h->ev_pool.empty()) {
    h->ev_pool.resize(8192);
    h->ev_kernel.assign(4096, (unsigned char)SSP_PROF_K_OTHER);
    for (auto& e : h->ev_pool) HIPCHK(hipEventCreate(&e));
  }
  return 0;
}

int ssp_profile_pause(ssp_handle* h, int paused) {
  if (!h) return fail(-1, "null handle");
  h->prof_paused = paused != 0;
  return 0;
}

int ssp_profile_read_kernel(ssp_handle* h, int kernel, double* ms, int64_t* launches, double* flops, double* executed_flops,
                            double* bytes) {
  if (!h) return fail(-1, "null handle");
  if (kernel < 0 || kernel >= SSP_PROF_K_COUNT) return fail(-1, "profile kernel bucket out of range");
  double tot = 0;
  for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
    if (h->ev_kernel[i / 2] != kernel) continue;
    float t = 0;
    HIPCHK(hipEventSynchronize(h->ev_pool[i + 1]));
    HIPCHK(hipEventElapsedTime(&t, h->ev_pool[i], h->ev_pool[i + 1]));
    tot += t;
  }
  const ssp_handle::ProfKernel& k = h->prof_k[kernel];
  if (ms) *ms = tot;
  if (launches) *launches = k.launches;
  if (flops) *flops = k.flops;
  if (executed_flops) *executed_flops = k.exec_flops;
  if (bytes) *bytes = k.bytes;
  return 0;
}

int ssp_profile_read(ssp_handle* h, double* ms, int64_t* launches, double* flops, double* bytes) {
  if (!h) return fail(-1, "null handle");
  double tot = 0;
  for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
    float t = 0;
    HIPCHK(hipEventSynchronize(h->ev_pool[i + 1]));
    HIPCHK(hipEventElapsedTime(&t, h->ev_pool[i], h->ev_pool[i + 1]));
    tot += t;
  }
  if (ms) *ms = tot;
  if (launches) *launches = h->prof_launches;
  if (flops) *flops = h->prof_flops;
  if (bytes) *bytes = h->prof_bytes;
  return 0;
}

int ssp_profile_read_executed(ssp_handle* h, double* executed_flops) {
  if (!h || !executed_flops) return fail(-1, "null handle / pointer");
  *executed_flops = h->prof_exec_flops;
  return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// forward / backward of one activation slot
// ------------------------------------------------------------------------------------------------
static const float* P(const ssp_handle* h, size_t off) { return h->buf.params_dev + off; }
static float* Gd(const ssp_handle* h, size_t off) { return h->buf.grads_dev + off; }

// BatchNorm statistics -> affine of layer l for every view of the set, ONE launch (view 0 then view 1 in the same thread)
static void bn_layer_args(ssp_handle* h, Slot* const* slots, int nviews, int l, double count, BnLayer* b) {
  const LayerDesc& d = h->L[l];
  for (int k = 0; k < nviews; ++k) {
    Slot& S = *slots[k];
    b[k].stats = S.bn[l].stats; b[k].gamma = P(h, d.g_off); b[k].beta = P(h, d.be_off);
    b[k].running_mean = h->buf.bn_running_dev + d.bn_ch_off;
    b[k].running_var = h->buf.bn_running_dev + h->n_bn_ch + d.bn_ch_off;
    b[k].scale = S.bn[l].scale; b[k].shift = S.bn[l].shift; b[k].mean = S.bn[l].mean; b[k].invstd = S.bn[l].invstd;
    b[k].C = d.cout; b[k].count = count;
  }
  if (nviews == 1) b[1] = b[0];
}
static int64_t* bn_nbt(ssp_handle* h, int l) {
  return h->buf.num_batches_tracked_dev ? h->buf.num_batches_tracked_dev + h->L[l].bn_index : nullptr;
}
static int bn_finalize(ssp_handle* h, Slot* const* slots, int nviews, int l, double count, int train, hipStream_t st) {
  BnLayer b[2];
  bn_layer_args(h, slots, nviews, l, count, b);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(h->L[l].cout * 32, 256)), dim3(256), 0, st, b[0], b[1], nviews, train, bn_nbt(h, l));
  HIPCHK(hipGetLastError());
  return 0;
}
// several layers whose statistics are complete at the same point: one finalize launch (blockIdx.y = layer)
static int bn_finalize_n(ssp_handle* h, Slot* const* slots, int nviews, const int* layers, int n, double count, int train, hipStream_t st) {
  if (n == 0) return 0;
  if (n == 1) return bn_finalize(h, slots, nviews, layers[0], count, train, st);
  if (n > 3) return fail(-3, "bn_finalize_n: at most three layers per launch");
  BnFinJobs J;
  J.n = n;
  int maxc = 0;
  for (int i = 0; i < n; ++i) {
    BnLayer b[2];
    bn_layer_args(h, slots, nviews, layers[i], count, b);
    J.L0[i] = b[0]; J.L1[i] = b[1]; J.nbt[i] = bn_nbt(h, layers[i]);
    maxc = std::max(maxc, h->L[layers[i]].cout);
  }
  hipLaunchKernelGGL(bn_finalize_multi_kernel, dim3(cdiv(maxc * 32, 256), n), dim3(256), 0, st, J, nviews, train);
  HIPCHK(hipGetLastError());
  return 0;
}
// conv_layer_fwd / conv_layer_fwd_bf16 with `deferred` != nullptr leave the finalize of their layer to the caller (bn_finalize_n)
struct BnDeferred {
  int layers[3];
  int n = 0;
};

static int pack_all(ssp_handle* h, bool with_bwd, int nprob, int N, int H, int W, hipStream_t st) {
  // default algorithm: every Winograd image (3x3 layers forward + data gradient, concatenated heads) in one launch
  const bool multi = pipe_algo() || g_conv_algo == 5 || g_conv_algo == 6;
  PackJobs J;
  J.n = 0;
  int nblocks = 0;
  auto add_job = [&](const float* w, float* dst, int cout_w, int cin_w, int tf, int nchunks_total, int chunk_off, int cob_off,
                     int ncob, int nchunks, bool w4 = false) {
    PackJob& q = J.j[J.n++];
    q.w = w; q.dst = dst; q.cout_w = cout_w; q.cin_w = cin_w; q.tf = tf; q.nchunks_total = nchunks_total;
    q.chunk_off = chunk_off; q.cob_off = cob_off; q.ncob = ncob; q.nchunks = nchunks; q.block0 = nblocks; q.w4 = w4 ? 1 : 0;
    nblocks += cdiv((long)ncob * nchunks * PK * NB, 256);  // one thread per (channel pair) cell: all components of a filter
  };
  auto pack = [&](const float* w, float* dst, int cout_w, int cin_w, int ks, int tf, bool wino, bool w4) -> int {
    const bool multi_now = pipe_algo() || g_conv_algo == 5 || g_conv_algo == 6;  // (the forward images of mode 8 are packed as algorithm 1)
    if (multi_now && wino && J.n < PACK_MAX_JOBS) {
      const int conv_cin = tf ? cout_w : cin_w, conv_cout = tf ? cin_w : cout_w;
      const int nchunks = 2 * cdiv(conv_cin, CK), ncob = cdiv(conv_cout, NB);
      add_job(w, dst, cout_w, cin_w, tf, nchunks, 0, 0, ncob, nchunks, w4);
      return 0;
    }
    return launch_pack(w, dst, cout_w, cin_w, ks, tf, wino, st, w4);
  };
  G1PackJobs G;
  G.n = 0;
  int g1_blocks = 0;
  PackBQueue BQ;   // (bf16 path)
  for (int l = 1; l < h->nlayers; ++l) {
    const LayerDesc& d = h->L[l];
    // the encoder's 3x3 layers at their resolution: the same F(4x4,3x3) predicate as the launches (conv_uses_w4)
    int lh = H / 8, lw = W / 8;
    if (l < 8) layer_res(l, H, W, lh, lw);
    h->pk_g1[l] = false;
    if (d.ks == 1 && !bf16_path() && g1_enabled() && G.n + 2 <= G1_PACK_MAX_JOBS) {
      const long npx = (long)N * lh * lw;
      const Slot& S0 = h->slot[0];
      const int src = l - 1, hcs = 256 * h->nheads;  // Pb <- Pa, Db <- Da, Sout <- DS: the layer before it in the table
      if (g1_fits(d.cin, d.cout, npx, S0.y_cs[src], S0.y_cs[l], nprob) &&
          g1_fits(d.cout, d.cin, npx, std::max(S0.y_cs[l], 80), hcs, nprob)) {
        h->pk_g1[l] = true;
        g1_add_pack(G, g1_blocks, P(h, d.w_off), h->wpk_g1_fwd[l], d.cout, d.cin, 0);
        if (with_bwd) g1_add_pack(G, g1_blocks, P(h, d.w_off), h->wpk_g1_bwd[l], d.cout, d.cin, 1);
        continue;
      }
    }
    if (bf16_path()) {  // bf16 operand images of conv_bf16_kernel (forward; mirrored / transposed for the data gradient)
      h->pk_w4_fwd[l] = h->pk_w4_bwd[l] = false; h->pk_g1[l] = false;
      CHK(BQ.add(P(h, d.w_off), reinterpret_cast<uint16_t*>(h->wpk_fwd + d.pk_fwd), d.cout, d.cin, d.ks, 0, 0, 0, st));
      if (with_bwd && (l < 8 || d.ks == 1))  // (the 3x3 heads share ONE concatenated data-gradient image, below)
        CHK(BQ.add(P(h, d.w_off), reinterpret_cast<uint16_t*>(h->wpk_bwd + d.pk_bwd), d.cout, d.cin, d.ks, 1, 0, 0, st));
      continue;
    }
    const bool wf = wino_ok(d.ks, d.cin), wb = wino_ok(d.ks, d.cout);
    {
      FwdAlgoScope fwd;
      h->pk_w4_fwd[l] = wf && d.ks == 3 && w4_eligible(h, nprob, N, lh, lw, d.cin, d.cout);
      CHK(pack(P(h, d.w_off), h->wpk_fwd + d.pk_fwd, d.cout, d.cin, d.ks, 0, wf, h->pk_w4_fwd[l]));
    }
    if (with_bwd) {
      h->pk_w4_bwd[l] = wb && d.ks == 3 && w4_eligible(h, nprob, N, lh, lw, (int)align_up(d.cout, 4), d.cin);
      CHK(pack(P(h, d.w_off), h->wpk_bwd + d.pk_bwd, d.cout, d.cin, d.ks, 1, wb, h->pk_w4_bwd[l]));
    }
  }
  h->packed_algo = g_conv_algo;
  h->packed_bwd = with_bwd;
  if (with_bwd && bf16_path()) {  // the same concatenation as a bf16 image: 8 chunks of 32 dY channels per head
    const int heads[3] = {L_PA, L_DA, L_DS};
    for (int k = 0; k < h->nheads; ++k)
      CHK(BQ.add(P(h, h->L[heads[k]].w_off), reinterpret_cast<uint16_t*>(h->wpk_heads_bwd), 256, 128, 3, 1, 8 * h->nheads, 8 * k, st));
  } else if (with_bwd) {  // concatenated data-gradient weights of the 3x3 heads: input channels = [Pa | Da | DS] dY
    const int heads[3] = {L_PA, L_DA, L_DS};
    const bool wino = wino_ok(3, 256 * h->nheads);
    const int total = 2 * 16 * (wino ? WC : 9) * CK * NB;
    for (int k = 0; k < h->nheads; ++k) {
#if SSP_LEGACY_ALGOS
      if (wino && bf16_algo())
        hipLaunchKernelGGL(pack_weights_wino8_bf16_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st,
                           P(h, h->L[heads[k]].w_off), reinterpret_cast<__bf16*>(h->wpk_heads_bwd), 256, 128, 1,
                           32 * h->nheads, 32 * k, 0, 2, 32, bf16_parts(true));
      else
#endif
      if (wino && multi && J.n < PACK_MAX_JOBS)
        add_job(P(h, h->L[heads[k]].w_off), h->wpk_heads_bwd, 256, 128, 1, 32 * h->nheads, 32 * k, 0, 2, 32);
      else if (wino && multi)
        hipLaunchKernelGGL(pack_weights_wino8_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st,
                           P(h, h->L[heads[k]].w_off), h->wpk_heads_bwd, 256, 128, 1, 32 * h->nheads, 32 * k, 0, 2, 32);
      else if (wino)
        hipLaunchKernelGGL(pack_weights_wino_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st,
                           P(h, h->L[heads[k]].w_off), h->wpk_heads_bwd, 256, 128, 1, 16 * h->nheads, 16 * k, 0, 2, 16);
      else
        hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, P(h, h->L[heads[k]].w_off),
                           h->wpk_heads_bwd, 256, 128, 3, 1, 16 * h->nheads, 16 * k, 0, 2, 16);
    }
    HIPCHK(hipGetLastError());
  }
  CHK(BQ.flush(st));   // every bf16 operand image of the step: one launch
  if (J.n > 0) {
    hipLaunchKernelGGL(pack_weights_wino8_multi_kernel, dim3(nblocks), dim3(256), 0, st, J);
    HIPCHK(hipGetLastError());
  }
  if (G.n > 0) {
    hipLaunchKernelGGL(pack_g1_kernel, dim3(g1_blocks), dim3(256), 0, st, G);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

// side stream + events of the handle (fork / join by events; inside a stream capture they become graph dependencies)
static int ensure_aux_stream(ssp_handle* h) {
  if (h->aux_stream != nullptr) return 0;
  HIPCHK(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&h->aux2_stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&h->ev_pack_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&h->ev_pack_join, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&h->ev_early_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&h->ev_early_join, hipEventDisableTiming));
  return 0;
}

// Fork / join of the handle's side stream: the destructor records the join event on the side stream and makes `st` wait for it
// unless join() already did - so an error return between fork and join neither leaves side-stream kernels racing with later
// work on `st` nor ends a stream capture with an unjoined fork (which would mask the original error).
struct SideFork {
  hipStream_t st, side; hipEvent_t ev_join; bool open;
  SideFork() : st(nullptr), side(nullptr), ev_join(nullptr), open(false) {}
  int fork(hipStream_t st_, hipStream_t side_, hipEvent_t ev_fork, hipEvent_t ev_join_) {
    st = st_; side = side_; ev_join = ev_join_;
    HIPCHK(hipEventRecord(ev_fork, st));
    HIPCHK(hipStreamWaitEvent(side, ev_fork, 0));
    open = true;
    return 0;
  }
  int join() {
    if (!open) return 0;
    open = false;
    HIPCHK(hipEventRecord(ev_join, side));
    HIPCHK(hipStreamWaitEvent(st, ev_join, 0));
    return 0;
  }
  ~SideFork() { (void)join(); }
};

// One or two activation slots processed together: the two views of a pair are independent problems of identical
// shape that share the weights, so the MFMA kernels take both in ONE launch (conv: XCDs 0-3 / 4-7; wgrad: one
// gradient accumulated over both) while the BatchNorm statistics stay per view (Train_model_heatmap_all.py:258,262).
struct SlotSet {
  int n;
  Slot* s[2];
};

// a layer whose finalize was left to its reader (ssp_handle::fin_pending) and whose reader cannot do it: the launch after all
static int bn_finalize_pending(ssp_handle* h, const SlotSet& SS, int l, int train, hipStream_t st) {
  if (l < 0 || !h->fin_pending[l]) return 0;
  h->fin_pending[l] = false;
  return bn_finalize(h, SS.s, SS.n, l, h->fin_count[l], train, st);
}

// bf16 path (conv algorithm 12): layer l on conv_bf16_kernel.  Every 3x3 layer writes a bf16 tensor (encoder: Y[l] and the raw
// pooled copy Apool[l]; heads: a slice of the [cells][256 heads] tensor) in the slot's buffers; the pointwise heads read those
// under BatchNorm + ReLU and write the fp32 logits / descriptors the loss kernels take.
static int conv_layer_fwd_bf16(ssp_handle* h, const SlotSet& SS, int l, int src, int N, int H, int W, int in_mode, int train,
                               hipStream_t st, BnDeferred* deferred = nullptr) {
  const LayerDesc& d = h->L[l];
  const bool pooled = in_mode == 2;  // input = raw pooled y of layer src (written by its conv), BatchNorm + ReLU on load
  ConvBCall c;
  c.nviews = SS.n; c.N = N; c.H = H; c.W = W; c.ks = d.ks; c.in_mode = 1;
  c.in_cs = pooled ? d.cin : SS.s[0]->y_cs[src]; c.in_co = pooled ? 0 : SS.s[0]->y_co[src]; c.cin = d.cin;
  c.wpk = reinterpret_cast<const uint16_t*>(h->wpk_fwd + d.pk_fwd); c.bias = P(h, d.b_off);
  c.out_cs = SS.s[0]->y_cs[l]; c.out_co = SS.s[0]->y_co[l]; c.cout = d.cout;
  c.out_f32 = d.ks == 1;
  const bool pool_out = l < 8 && SS.s[0]->Apool[l] != nullptr && d.bn;
  // 3x3 layers on the 30x40 maps (layers 6, 7 and the three heads): the input is activated once into act[src]
  // (bn_relu_bf16_kernel: 20 MB, 9 us) and read without staging arithmetic by every unit and by the weight gradient
  static const int act_env = getenv("SSP_ACT7") ? atoi(getenv("SSP_ACT7")) : 1;   // (perf-debug A/B: 0 off, 1 heads only, 2 also layers 6, 7)
  const bool from_act = d.ks == 3 && d.cin == 128 && src >= 5 && src <= 7 && (src == 7 ? act_env != 0 : act_env >= 2) &&
                        SS.s[0]->act[src] != nullptr && (pooled ? SS.s[0]->pool_raw[src] : (SS.s[0]->y_cs[src] == 128 && SS.s[0]->y_co[src] == 0));
  if (from_act && !SS.s[0]->act_valid[src]) {
    Slot &S0 = *SS.s[0], &S1 = *SS.s[SS.n - 1];
    const long nitems = (long)N * H * W * 16;
    const float* y0 = pooled ? S0.Apool[src] : S0.Y[src];
    const float* y1 = pooled ? S1.Apool[src] : S1.Y[src];
    hipLaunchKernelGGL(bn_relu_bf16_kernel, dim3((unsigned)std::min<long>(cdiv(nitems, 256), 4096), SS.n), dim3(256), 0, st,
                       reinterpret_cast<const uint16_t*>(y0), reinterpret_cast<const uint16_t*>(y1), S0.bn[src].scale, S0.bn[src].shift,
                       S1.bn[src].scale, S1.bn[src].shift, reinterpret_cast<uint16_t*>(S0.act[src]), reinterpret_cast<uint16_t*>(S1.act[src]),
                       nitems, 128);
    HIPCHK(hipGetLastError());
    for (int k = 0; k < SS.n; ++k) SS.s[k]->act_valid[src] = true;
  }
  const bool from_act7 = from_act;
  if (from_act) c.in_mode = 0;
  for (int k = 0; k < SS.n; ++k) {
    Slot& S = *SS.s[k];
    if (pooled && !S.pool_raw[src]) return fail(-3, "bf16 path: layer %d has no raw pooled output", src);
    c.in[k] = from_act7 ? S.act[src] : pooled ? S.Apool[src] : S.Y[src];
    c.out[k] = S.Y[l];
    c.in_scale[k] = S.bn[src].scale; c.in_shift[k] = S.bn[src].shift;
    c.stats[k] = (d.bn && train) ? S.bn[l].stats : nullptr;
    c.pool_out[k] = pool_out ? reinterpret_cast<uint16_t*>(S.Apool[l]) : nullptr;
    if (l < 8) S.pool_raw[l] = pool_out;
  }
  if (pool_out) c.pool_gamma = P(h, d.g_off);
  {
    const double flops = 2.0 * SS.n * N * H * W * (double)d.cin * d.cout * d.ks * d.ks;
    const double bytes = SS.n * (double)N * H * W * (2.0 * d.cin + (c.out_f32 ? 4.0 : 2.0) * d.cout + (pool_out ? 0.5 * d.cout : 0.0));  // (+ the raw pooled copy)
    ProfScope ps(h, d.ks == 3 ? SSP_PROF_CONV3X3_FWD : -1, st, flops, bytes, flops, SSP_PROF_K_CONV_BF16);
    CHK(launch_conv_bf16(c, h->n_cu, st));
  }
  if (d.bn && deferred != nullptr && deferred->n < 3) deferred->layers[deferred->n++] = l;
  else if (d.bn) CHK(bn_finalize(h, SS.s, SS.n, l, (double)N * H * W, train, st));
  return 0;
}

static int conv_layer_fwd(ssp_handle* h, const SlotSet& SS, int l, int src, int N, int H, int W, int in_mode, int train,
                          hipStream_t st, BnDeferred* deferred = nullptr) {
  const LayerDesc& d = h->L[l];
  Slot& A = *SS.s[0];
  if (bf16_path()) return conv_layer_fwd_bf16(h, SS, l, src, N, H, W, in_mode, train, st, deferred);
  FwdAlgoScope fwd_algo;
  const bool pooled = in_mode == 2;  // input = pooled output of layer src: raw pooled y (BatchNorm + ReLU on load, mode 1)
                                     // when its conv wrote it (pool_raw), else materialised maxpool(relu(bn(Y_src))) (mode 0)
  if (pooled && A.pool_raw[src]) {
    in_mode = 1;
  } else if (pooled) {
    CHK(bn_finalize_pending(h, SS, src, train, st));   // (bn_relu_pool_kernel reads the affine)
    {
      Slot &S0 = *SS.s[0], &S1 = *SS.s[SS.n - 1];
      const long total = (long)N * H * W * (d.cin / 4);
      hipLaunchKernelGGL(bn_relu_pool_kernel, dim3(std::min(cdiv(total, 256), 8192), SS.n), dim3(256), 0, st, S0.Y[src],
                         S0.bn[src].scale, S0.bn[src].shift, S0.Apool[src], S1.Y[src], S1.bn[src].scale, S1.bn[src].shift,
                         S1.Apool[src], N, 2 * H, 2 * W, d.cin);
    }
    HIPCHK(hipGetLastError());
    in_mode = 0;
  }
  ConvCall c;
  c.in = pooled ? A.Apool[src] : A.Y[src]; c.in_cs = A.y_cs[src]; c.in_co = A.y_co[src]; c.cin = d.cin;
  c.wpk = h->wpk_fwd + d.pk_fwd; c.bias = P(h, d.b_off); c.wino = wino_ok(d.ks, d.cin);
  c.out = A.Y[l]; c.out_cs = A.y_cs[l]; c.out_co = A.y_co[l]; c.cout = d.cout;
  c.in_scale = A.bn[src].scale; c.in_shift = A.bn[src].shift;
  c.stats = (d.bn && train) ? A.bn[l].stats : nullptr;
  c.N = N; c.H = H; c.W = W; c.ks = d.ks; c.in_mode = in_mode; c.nchunks = d.nchunks_fwd; c.ncob = d.ncob_fwd;
  c.force_w4 = h->pk_w4_fwd[l] ? 1 : 0;
  if (SS.n == 2) {
    Slot& B = *SS.s[1];
    c.nprob = 2; c.in2 = pooled ? B.Apool[src] : B.Y[src]; c.out2 = B.Y[l]; c.in_scale2 = B.bn[src].scale;
    c.in_shift2 = B.bn[src].shift;
    c.stats2 = (d.bn && train) ? B.bn[l].stats : nullptr;
  }
  // layers followed by BatchNorm + ReLU + MaxPool (1, 3, 5): the conv itself writes the raw pooled copy its consumers read
  if (l < 8) {
    const bool raw = A.Apool[l] != nullptr && d.bn && conv_writes_pool(h, c);
    if (raw) {
      c.pool_out[0] = A.Apool[l]; c.pool_out[1] = SS.n == 2 ? SS.s[1]->Apool[l] : nullptr; c.pool_gamma = P(h, d.g_off);
    }
    for (int k = 0; k < SS.n; ++k) SS.s[k]->pool_raw[l] = raw;
  }
  // the input layer's statistics are still raw (fin_pending): this launch derives the affine itself where its kernel can
  if (src >= 0 && h->fin_pending[src]) {
    if (conv_has_bn_lazy(h, c)) {
      const LayerDesc& ds = h->L[src];
      BnLazy& z = c.lazy;
      for (int k = 0; k < SS.n; ++k) {
        Slot& S = *SS.s[k];
        z.stats[k] = S.bn[src].stats; z.scale[k] = S.bn[src].scale; z.shift[k] = S.bn[src].shift; z.mean[k] = S.bn[src].mean;
        z.invstd[k] = S.bn[src].invstd;
      }
      if (SS.n == 1) { z.stats[1] = z.stats[0]; z.scale[1] = z.scale[0]; z.shift[1] = z.shift[0]; z.mean[1] = z.mean[0]; z.invstd[1] = z.invstd[0]; }
      z.gamma = P(h, ds.g_off); z.beta = P(h, ds.be_off);
      z.running_mean = h->buf.bn_running_dev + ds.bn_ch_off;
      z.running_var = h->buf.bn_running_dev + h->n_bn_ch + ds.bn_ch_off;
      z.nbt = bn_nbt(h, src); z.count = h->fin_count[src]; z.C = ds.cout; z.nviews = SS.n; z.mode = 2;
      h->fin_pending[src] = false;
    } else {
      CHK(bn_finalize_pending(h, SS, src, train, st));
    }
  }
  CHK(launch_conv(h, c, st, d.ks == 3 ? SSP_PROF_CONV3X3_FWD : 0));
  if (d.bn && deferred != nullptr && deferred->n < 3) deferred->layers[deferred->n++] = l;
  else if (d.bn && train && l < 8 && bn_lazy_env() && !bf16_algo()) {   // encoder layers: the next convolution finalizes (or bn_finalize_pending)
    h->fin_pending[l] = true;
    h->fin_count[l] = (double)N * H * W;
  }
  else if (d.bn) CHK(bn_finalize(h, SS.s, SS.n, l, (double)N * H * W, train, st));  // view 0 then 1 inside the kernel
  return 0;
}

static int run_forward(ssp_handle* h, const SlotSet& SS, const float* const* xs, int N, int H, int W, int train,
                       bool for_backward, hipStream_t st, bool detector_only = false, bool zero_ddesc = false) {
  for (int k = 0; k < SS.n; ++k) {
    Slot& S = *SS.s[k];
    S.N = N; S.H = H; S.W = W; S.x = xs[k];
    for (int l = 0; l < 8; ++l) S.act_valid[l] = false;   // (rewritten by this forward where a layer uses it)
    S.bsums_dirty = false;
  }
  if (SS.n == 2 && SS.s[0]->stats_bytes == SS.s[1]->stats_bytes) CHK(dev_zero2(SS.s[0]->stats_region, SS.s[1]->stats_region, SS.s[0]->stats_bytes, st));
  else for (int k = 0; k < SS.n; ++k) CHK(dev_zero(SS.s[k]->stats_region, SS.s[k]->stats_bytes, st));
  // The Winograd weight images (a ~95 us latency-bound launch over 2.6 M floats) are packed on the side stream beside the
  // HBM-bound first-layer convolution, which needs none of them; layer 1 waits for the join.
  static const int pack_stream_env = getenv("SSP_PACK_STREAM") ? atoi(getenv("SSP_PACK_STREAM")) : 1;  // (perf-debug: 0 = in line)
  const bool pack_forked = pack_stream_env != 0;
  SideFork pack_fork;
  if (pack_forked) {
    CHK(ensure_aux_stream(h));
    CHK(pack_fork.fork(st, h->aux_stream, h->ev_pack_fork, h->ev_pack_join));
    CHK(pack_all(h, for_backward, SS.n, N, H, W, h->aux_stream));
  } else {
    CHK(pack_all(h, for_backward, SS.n, N, H, W, st));
  }
  // layer 0: direct 1->64 conv (HBM-bound; the views ride one launch, blockIdx.y)
  {
    const LayerDesc& d = h->L[0];
    Slot &S0 = *SS.s[0], &S1 = *SS.s[SS.n - 1];
    if (bf16_path())
      hipLaunchKernelGGL(conv0_direct_kernel<uint16_t>, dim3(l0_resident_grid(conv0_direct_kernel<uint16_t>, h, SS.n, (long)N * H, W), SS.n), dim3(256), l0_lds_bytes(W), st, S0.x, S1.x, P(h, d.w_off),
                         P(h, d.b_off), reinterpret_cast<uint16_t*>(S0.Y[0]), reinterpret_cast<uint16_t*>(S1.Y[0]),
                         train ? S0.bn[0].stats : nullptr, train ? S1.bn[0].stats : nullptr, N, H, W);
    else
    hipLaunchKernelGGL(conv0_direct_kernel<float>, dim3(l0_resident_grid(conv0_direct_kernel<float>, h, SS.n, (long)N * H, W), SS.n), dim3(256), l0_lds_bytes(W), st, S0.x, S1.x, P(h, d.w_off),
                       P(h, d.b_off), S0.Y[0], S1.Y[0], train ? S0.bn[0].stats : nullptr, train ? S1.bn[0].stats : nullptr,
                       N, H, W);
    HIPCHK(hipGetLastError());
    if (train && bn_lazy_env() && !bf16_path()) { h->fin_pending[0] = true; h->fin_count[0] = (double)N * H * W; }   // layer 1 finalizes
    else CHK(bn_finalize(h, SS.s, SS.n, 0, (double)N * H * W, train, st));
  }
  CHK(pack_fork.join());
  for (int l = 1; l < 8; ++l) {
    int lh, lw; layer_res(l, H, W, lh, lw);
    CHK(conv_layer_fwd(h, SS, l, l - 1, N, lh, lw, layer_in_mode(l), train, st));
  }
  const int Hc = H / 8, Wc = W / 8;
  // the pointwise layers Pb, Db, Sout of both views: ONE grouped launch after the 3x3 head convs (conv1x1_group_kernel) when
  // pack_all wrote their images, else one conv_mfma_kernel<1, ...> launch each
  auto pointwise = [&](const int* layers, int n) -> int {
    bool grouped = true;
    for (int i = 0; i < n; ++i) grouped = grouped && h->pk_g1[layers[i]];
    if (!grouped) {   // (the bf16 path: one launch per head, their BatchNorm finalizes together behind the last one)
      BnDeferred fin;
      for (int i = 0; i < n; ++i) CHK(conv_layer_fwd(h, SS, layers[i], layers[i] - 1, N, Hc, Wc, 1, train, st, &fin));
      return bn_finalize_n(h, SS.s, SS.n, fin.layers, fin.n, (double)N * Hc * Wc, train, st);
    }
    G1Layer Lg[3];
    for (int i = 0; i < n; ++i) {
      const int l = layers[i], src = l - 1;
      const LayerDesc& d = h->L[l];
      G1Layer& y = Lg[i];
      y.in_cs = SS.s[0]->y_cs[src]; y.in_co = SS.s[0]->y_co[src]; y.out_cs = SS.s[0]->y_cs[l]; y.out_co = SS.s[0]->y_co[l];
      y.wpk = h->wpk_g1_fwd[l]; y.bias = P(h, d.b_off); y.K = d.cin; y.N = d.cout;
      for (int k = 0; k < SS.n; ++k) {
        Slot& S = *SS.s[k];
        y.in[k] = S.Y[src]; y.out[k] = S.Y[l]; y.scale[k] = S.bn[src].scale; y.shift[k] = S.bn[src].shift;
        y.stats[k] = (d.bn && train) ? S.bn[l].stats : nullptr;
      }
    }
    CHK(launch_g1(Lg, n, SS.n, (long)N * Hc * Wc, 1, h->n_cu, st));
    int bl[3], nbl = 0;
    for (int i = 0; i < n; ++i)
      if (h->L[layers[i]].bn) bl[nbl++] = layers[i];
    return bn_finalize_n(h, SS.s, SS.n, bl, nbl, (double)N * Hc * Wc, train, st);   // convDb + convPb: one launch
  };
  const int pw[3] = {L_DB, L_SOUT, L_PB};
  BnDeferred fin3;   // the 3x3 heads all read layer 7: their launches back to back, ONE finalize launch for their statistics
  CHK(conv_layer_fwd(h, SS, L_PA, 7, N, Hc, Wc, 1, train, st, &fin3));
  if (detector_only) {
    CHK(bn_finalize_n(h, SS.s, SS.n, fin3.layers, fin3.n, (double)N * Hc * Wc, train, st));
    return pointwise(pw + 2, 1);
  }
  CHK(conv_layer_fwd(h, SS, L_DA, 7, N, Hc, Wc, 1, train, st, &fin3));
  if (h->nheads == 3) CHK(conv_layer_fwd(h, SS, L_DS, 7, N, Hc, Wc, 1, train, st, &fin3));
  CHK(bn_finalize_n(h, SS.s, SS.n, fin3.layers, fin3.n, (double)N * Hc * Wc, train, st));
  if (h->nheads == 3) {
    CHK(pointwise(pw, 3));
  } else {
    const int pw2[2] = {L_DB, L_PB};
    CHK(pointwise(pw2, 2));
  }
  for (int l = 0; l < 8; ++l) CHK(bn_finalize_pending(h, SS, l, train, st));   // (none is left on the shipped layer table)
  const int ncells = N * Hc * Wc;
  {
    Slot &S = *SS.s[0], &T = *SS.s[SS.n - 1];   // both views in one launch (blockIdx.y)
    hipLaunchKernelGGL(desc_normalize_kernel, dim3(cdiv(ncells, 4), SS.n), dim3(256), 0, st, S.Y[L_DB], S.bn[L_DB].scale,
                       S.bn[L_DB].shift, S.desc, S.inv_norm, zero_ddesc ? S.ddesc : (float*)nullptr, ncells, S.y_cs[L_DB], S.y_co[L_DB],
                       T.Y[L_DB], T.bn[L_DB].scale, T.bn[L_DB].shift, T.desc, T.inv_norm, zero_ddesc ? T.ddesc : (float*)nullptr);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// BatchNorm(+ReLU(+pool)) backward of layer l for every view of the set in the same launches: dout[k] -> dy[k]
// (+ dgamma, dbeta, conv-bias gradient, accumulated over the views)
static int bn_layer_backward(ssp_handle* h, const SlotSet& SS, int l, const float* const* dout, int d_cs, int d_co, bool relu,
                             bool pool_after, float* const* dy, int dy_cs, int dy_co, int N, int H, int W, hipStream_t st,
                             BnSumsQueue* queue = nullptr, BnPlainJob* collect = nullptr) {
  const LayerDesc& d = h->L[l];
  BnBwdArgs a[2];
  bool have_pool = true;
  for (int k = 0; k < SS.n; ++k) {
    Slot& S = *SS.s[k];
    BnBwdArgs& v = a[k];
    v.y = S.Y[l]; v.dout = dout[k]; v.dy = dy[k]; v.scale = S.bn[l].scale; v.shift = S.bn[l].shift; v.mean = S.bn[l].mean;
    v.invstd = S.bn[l].invstd; v.gamma = P(h, d.g_off); v.sums = S.bn[l].bsums; v.dbias = Gd(h, d.b_off);
    v.x = S.x; v.apool = l < 8 ? S.Apool[l] : nullptr; v.beta = P(h, d.be_off); v.pool_fix = 0;
    v.N = N; v.H = H; v.W = W; v.C = d.cout; v.y_cs = S.y_cs[l]; v.y_co = S.y_co[l]; v.d_cs = d_cs; v.d_co = d_co;
    v.dy_cs = dy_cs; v.dy_co = dy_co; v.count = (double)N * H * W;
    v.k12 = S.bn[l].k12;
    have_pool = have_pool && v.apool != nullptr;
  }
  const bool fused = h->bsums_fused[l];  // pass 1 already sits in bsums (conv_layer_backward of the layer above)
  h->bsums_fused[l] = false;
  h->apply_fused[l] = false;             // set below only by the branches that leave pass 2 to the weight gradient
  const bool raw_pool = pool_after && l < 8 && SS.s[0]->pool_raw[l];
  if ((fused || raw_pool) && pool_after)  // S2 of gamma == 0 channels comes from a scan over Y (bn_bwd_sums_kernel)
    for (int k = 0; k < SS.n; ++k) a[k].pool_fix = 1;
  const BnBwdArgs &a0 = a[0], &a1 = a[SS.n - 1];
  float *dg = Gd(h, d.g_off), *db = Gd(h, d.be_off);
  if (l == 0) {
    // pass 1 (sums), then pass 2 fused with the first layer's weight gradient (dY0 is never materialised);
    // both passes recompute Y0 from the image instead of reading S.Y[0]
    const int nb1 = l0_resident_grid(bn_bwd_reduce_l0_kernel<float>, h, SS.n, (long)N * H, W);
    const int nb2 = l0_resident_grid(bn_bwd_apply_l0_kernel<float>, h, SS.n, (long)N * H, W);
    if (!fused)  // else: S1 / S2 were accumulated by the data-gradient conv of layer 1 (setup_bnr)
      hipLaunchKernelGGL(bn_bwd_reduce_l0_kernel<float>, dim3(nb1, SS.n), dim3(256), l0_lds_bytes(W), st, a0, a1, P(h, d.w_off), P(h, d.b_off));
    hipLaunchKernelGGL(bn_bwd_sums_kernel<float>, dim3(cdiv(64 * 32, 256)), dim3(256), 0, st, a0, a1, SS.n, dg, db);
    hipLaunchKernelGGL(bn_bwd_apply_l0_kernel<float>, dim3(nb2, SS.n), dim3(256), l0_lds_bytes(W), st, a0, a1, P(h, d.w_off), P(h, d.b_off),
                       Gd(h, d.w_off));
  } else if (relu && pool_after && have_pool && d.cout % 4 == 0 && d_cs == d.cout && d_co == 0) {
    // pass 1 from the pooled activation (1/4 of Y's bytes), pass 2 over Y
    const long npix = (long)N * (H / 2) * (W / 2);
    const int rows = 256 / (d.cout / 4);
    const int nb = std::max(1, std::min(cdiv(npix, rows), 1024));
    if (!fused && !SS.s[0]->pool_raw[l]) hipLaunchKernelGGL(bn_bwd_reduce_pool_kernel, dim3(nb, SS.n), dim3(256), 0, st, a0, a1, P(h, d.be_off));
    else if (!fused) {
      // raw pooled y in Apool: the window's arg-max of z is that element, so pass 1 is the plain ReLU-layer reduction over the
      // quarter-size tensors (y = Apool, dOut)
      BnBwdArgs r[2] = {a[0], a[SS.n - 1]};
      for (int k = 0; k < 2; ++k) {
        r[k].y = r[k].apool; r[k].y_cs = d.cout; r[k].y_co = 0; r[k].H = H / 2; r[k].W = W / 2;
      }
      hipLaunchKernelGGL((bn_bwd_kernel<true, false, false>), dim3(nb, SS.n), dim3(256), 0, st, r[0], r[1], r[0], r[1]);
    }
    hipLaunchKernelGGL(bn_bwd_sums_kernel<float>, dim3(cdiv(d.cout * 32, 256)), dim3(256), 0, st, a0, a1, SS.n, dg, db);
    // pass 2: inside the layer's weight gradient (it stages dY anyway and writes it for the data gradient) where possible
    // (the pooled layers 1, 3, 5 read the un-pooled activation of layers 0, 2, 4: input mode 1)
    const bool defer = l >= 1 && l < 8 && layer_in_mode(l) == 1 && dy_cs == d.cout && dy_co == 0 && SS.s[0]->y_cs[l] == d.cout &&
                       SS.s[0]->y_co[l] == 0 && wgrad_can_fuse_apply(d.ks, 1, H, W, d.cout);
    h->apply_fused[l] = defer;
    if (!defer) hipLaunchKernelGGL((bn_bwd_kernel<true, true, true>), dim3(nb, SS.n), dim3(256), 0, st, a0, a1, a0, a1);
  } else if (relu && pool_after) CHK((launch_bn_bwd<true, true>(a, SS.n, dg, db, st, fused)));
  else if (relu) {
    // encoder layers 2, 4, 6, 7 (dense [N,H,W,C] tensors) and the 3x3 heads (slices of [cells][256 heads] tensors: y, the
    // gradient and dY share channel stride and offset): pass 2 inside the weight gradient as well
    const bool defer = l >= 1 && d_cs == SS.s[0]->y_cs[l] && d_co == SS.s[0]->y_co[l] && dy_cs == d_cs && dy_co == d_co &&
                       wgrad_can_fuse_apply(d.ks, 1, H, W, d.cout);
    h->apply_fused[l] = defer;
    // pass 1 came from the data gradient above, pass 2 goes into the weight gradient: what is left is the replica reduction, and
    // the fused weight gradient can do that in its prologue (WgradArgs::f_lazy) - no launch at all
    if (fused && defer && bn_lazy_env() && !bf16_algo()) h->sums_lazy[l] = true;
    else CHK((launch_bn_bwd<true, false>(a, SS.n, dg, db, st, fused, defer, queue)));
  }
  else if (collect != nullptr) {   // BatchNorm without ReLU (the pointwise heads): the caller launches two layers together
    collect->a[0] = a0; collect->a[1] = a1; collect->dg = dg; collect->db = db; collect->filled = true;
    return 0;
  }
  else CHK((launch_bn_bwd<false, false>(a, SS.n, dg, db, st)));
  HIPCHK(hipGetLastError());
  return 0;
}
// BatchNorm backward of TWO plain layers (no ReLU, no pooling: convPb and convDb) in three launches instead of six
static int launch_bn_bwd_pair(const BnPlainJob& p, const BnPlainJob& q, int nviews, hipStream_t st) {
  auto blocks = [](const BnBwdArgs& a) {
    const int nq = (a.C + 3) / 4, rows = 256 / nq;
    return std::min(cdiv((long)a.N * a.H * a.W, rows), 1024);
  };
  const int nb = std::max(blocks(p.a[0]), blocks(q.a[0]));
  hipLaunchKernelGGL((bn_bwd_kernel<false, false, false, float>), dim3(nb, 4), dim3(256), 0, st, p.a[0], p.a[1], q.a[0], q.a[1]);
  BnSumsQueue sq;
  sq.J.n = 2; sq.maxC = std::max(p.a[0].C, q.a[0].C);
  sq.J.a0[0] = p.a[0]; sq.J.a1[0] = p.a[1]; sq.J.dgamma[0] = p.dg; sq.J.dbeta[0] = p.db;
  sq.J.a0[1] = q.a[0]; sq.J.a1[1] = q.a[1]; sq.J.dgamma[1] = q.dg; sq.J.dbeta[1] = q.db;
  CHK(flush_bn_sums(sq, nviews, st));
  hipLaunchKernelGGL((bn_bwd_kernel<false, false, true, float>), dim3(nb, 4), dim3(256), 0, st, p.a[0], p.a[1], q.a[0], q.a[1]);
  HIPCHK(hipGetLastError());
  return 0;
}

// Ask the data-gradient conv `c` (output = gradient wrt the activation of layer src) to accumulate pass 1 of layer src's
// BatchNorm backward in its epilogue; records the fact for bn_layer_backward(src).
static void setup_bnr(ssp_handle* h, const SlotSet& SS, int src, bool pooled, ConvCall& c) {
  // (src == 0 too: the first layer's pass 1 then reads Y0 inside the MFMA-bound data-gradient launch of layer 1 instead of
  // running bn_bwd_reduce_l0_kernel, an HBM-bound pass over dOut0 of its own)
  if (src < 0 || !h->L[src].bn || !can_fuse_bnr(c) || c.out_co != 0 || c.out_cs != c.cout || c.cout != h->L[src].cout) return;
  const LayerDesc& ds = h->L[src];
  for (int k = 0; k < SS.n; ++k) {
    Slot& S = *SS.s[k];
    if (pooled && S.pool_raw[src]) {
      // raw pooled y: the window's arg-max of z IS this element (max for gamma >= 0, min for gamma < 0), so the ReLU-layer
      // formulas apply to the quarter-size tensor (gamma == 0 channels: pool_fix in bn_layer_backward, as before)
      if (S.Apool[src] == nullptr) return;
      c.bnr_t[k] = S.Apool[src];
      c.bnr_p[0][k] = S.bn[src].scale; c.bnr_p[1][k] = S.bn[src].shift; c.bnr_p[2][k] = S.bn[src].mean;
      c.bnr_p[3][k] = S.bn[src].invstd;
    } else if (pooled) {
      if (S.Apool[src] == nullptr) return;
      c.bnr_t[k] = S.Apool[src];
      c.bnr_p[0][k] = P(h, ds.be_off); c.bnr_p[1][k] = P(h, ds.g_off);
    } else {
      c.bnr_t[k] = S.Y[src];
      c.bnr_p[0][k] = S.bn[src].scale; c.bnr_p[1][k] = S.bn[src].shift; c.bnr_p[2][k] = S.bn[src].mean;
      c.bnr_p[3][k] = S.bn[src].invstd;
    }
  }
  c.bnr_mode = (pooled && !SS.s[0]->pool_raw[src]) ? 2 : 1;
  c.bnr_cs = pooled ? ds.cout : SS.s[0]->y_cs[src];
  c.bnr_co = pooled ? 0 : SS.s[0]->y_co[src];
  if (c.bnr_cs % 4 != 0 || c.bnr_co % 4 != 0) { c.bnr_mode = 0; return; }
  c.stats = SS.s[0]->bn[src].bsums;
  if (SS.n == 2) c.stats2 = SS.s[1]->bn[src].bsums;
  h->bsums_fused[src] = true;
}

// weight gradient (both views in one launch) and data gradient (both views in one launch) of conv layer l, given
// dY of each view in dy[k] (channel stride dy_cs, offset dy_co); the data gradient goes to din[k].
static int conv_layer_backward(ssp_handle* h, const SlotSet& SS, int l, int src, float* const* dy, int dy_cs, int dy_co,
                               float* const* din, int din_cs, int din_co, int N, int H, int W, int in_mode,
                               hipStream_t st, bool skip_dgrad = false, bool skip_wgrad = false) {
  const LayerDesc& d = h->L[l];
  Slot& A = *SS.s[0];
  const bool pooled = in_mode == 2;  // Apool[src] holds the pooled input: raw pooled y (pool_raw: BatchNorm + ReLU on load)
                                     // or the materialised maxpool(relu(bn(Y_src)))
  if (pooled) in_mode = A.pool_raw[src] ? 1 : 0;
  WgradCall w;
  w.in = pooled ? A.Apool[src] : A.Y[src]; w.in_cs = A.y_cs[src]; w.in_co = A.y_co[src]; w.cin = d.cin;
  w.dout = dy[0]; w.dout_cs = dy_cs; w.dout_co = dy_co; w.cout = d.cout;
  w.in_scale = A.bn[src].scale; w.in_shift = A.bn[src].shift; w.dw = Gd(h, d.w_off);
  w.N = N; w.H = H; w.W = W; w.ks = d.ks; w.in_mode = in_mode;
  ConvCall c;
  c.in = dy[0]; c.in_cs = dy_cs; c.in_co = dy_co; c.cin = (int)align_up(d.cout, 4);
  c.wpk = h->wpk_bwd + d.pk_bwd; c.bias = nullptr; c.wino = wino_ok(d.ks, d.cout);
  c.out = din[0]; c.out_cs = din_cs; c.out_co = din_co; c.cout = d.cin;
  c.in_scale = nullptr; c.in_shift = nullptr; c.stats = nullptr; c.backward = true;
  c.N = N; c.H = H; c.W = W; c.ks = d.ks; c.in_mode = 0; c.nchunks = d.nchunks_bwd; c.ncob = d.ncob_bwd;
  c.force_w4 = h->pk_w4_bwd[l] ? 1 : 0;
  if (SS.n == 2) {
    Slot& B = *SS.s[1];
    w.nprob = 2; w.in2 = pooled ? B.Apool[src] : B.Y[src]; w.dout2 = dy[1]; w.in_scale2 = B.bn[src].scale;
    w.in_shift2 = B.bn[src].shift;
    c.nprob = 2; c.in2 = dy[1]; c.out2 = din[1];
  }
  if (d.ks == 3 && src < 8) setup_bnr(h, SS, src, pooled, c);  // layer src: BatchNorm + ReLU (+ pool) of the encoder
  if (l < 8 && h->apply_fused[l]) {
    // bn_layer_backward(l) ran the sums only: dOut of the pooled activation still sits in gP (= din, which the data-gradient
    // conv overwrites AFTER the weight gradient has consumed it: same stream), dY is produced into dy[] by the weight gradient
    h->apply_fused[l] = false;
    w.fuse_apply = true;
    w.fuse_pool = (l == 1 || l == 3 || l == 5);
    w.dout_cs = d.cout; w.dout_co = 0; w.f_gamma = P(h, d.g_off); w.f_ycs = A.y_cs[l];
    for (int k = 0; k < SS.n; ++k) {
      Slot& S = *SS.s[k];
      (k ? w.dout2 : w.dout) = din[k];
      w.f_y[k] = S.Y[l]; w.f_dy[k] = dy[k]; w.f_scale[k] = S.bn[l].scale; w.f_shift[k] = S.bn[l].shift;
      w.f_mean[k] = S.bn[l].mean; w.f_invstd[k] = S.bn[l].invstd; w.f_k12[k] = S.bn[l].k12;
    }
    if (h->sums_lazy[l]) {
      h->sums_lazy[l] = false;
      w.f_lazy = 1; w.f_count = (double)N * H * W; w.f_dgamma = Gd(h, d.g_off); w.f_dbeta = Gd(h, d.be_off); w.f_dbias = Gd(h, d.b_off);
      for (int k = 0; k < SS.n; ++k) w.f_bsums[k] = SS.s[k]->bn[l].bsums;
    }
  }
  if (!skip_wgrad) CHK(launch_wgrad(h, w, h->partial, h->partial_floats, h->n_cu, st));
  if (skip_dgrad) return 0;  // the caller runs the data gradient itself (the grouped pointwise launch of the heads)
  CHK(launch_conv(h, c, st, d.ks == 3 ? SSP_PROF_CONV3X3_DGRAD : 0));
  return 0;
}

// bf16 path (conv algorithm 12): encoder layers l_hi .. l_lo.  gP holds dOut of layer l_hi (bf16, grad wrt its (pooled) activation);
// per layer: BatchNorm + ReLU (+ pool) backward in fp32 arithmetic on the bf16 tensors (sums, then apply: dY -> gQ, bf16), weight
// gradient and data gradient on the bf16 matrix cores (-> gP, bf16).
static int encoder_backward_bf16(ssp_handle* h, const SlotSet& SS, int l_hi, int l_lo, hipStream_t st) {
  Slot& S0 = *SS.s[0];
  const int N = S0.N, H = S0.H, W = S0.W;
  for (int l = l_hi; l >= l_lo; --l) {
    const LayerDesc& d = h->L[l];
    int lh, lw; layer_res(l, H, W, lh, lw);
    const bool pool_after = (l == 1 || l == 3 || l == 5);
    const int C = d.cout;
    BnBwdArgs a[2];
    for (int k = 0; k < SS.n; ++k) {
      Slot& S = *SS.s[k];
      BnBwdArgs& v = a[k];
      v.y = S.Y[l]; v.dout = S.gP; v.dy = S.gQ; v.scale = S.bn[l].scale; v.shift = S.bn[l].shift; v.mean = S.bn[l].mean;
      v.invstd = S.bn[l].invstd; v.gamma = P(h, d.g_off); v.sums = S.bn[l].bsums; v.dbias = Gd(h, d.b_off);
      v.x = S.x; v.apool = nullptr; v.beta = P(h, d.be_off); v.pool_fix = 0;
      v.N = N; v.H = lh; v.W = lw; v.C = C; v.y_cs = C; v.y_co = 0; v.d_cs = C; v.d_co = 0; v.dy_cs = C; v.dy_co = 0;
      v.count = (double)N * lh * lw; v.k12 = S.bn[l].k12;
    }
    float *dg = Gd(h, d.g_off), *db = Gd(h, d.be_off);
    // pass 1 of this layer's BatchNorm backward already sits in bsums: the data-gradient launch of the layer above accumulated it
    // in its copy-out (ConvBArgs::bnr_*)
    const bool sums_fused = h->bsums_fused[l];
    h->bsums_fused[l] = false;
    if (l == 0) {
      const BnBwdArgs &a0 = a[0], &a1 = a[SS.n - 1];
      const int nb1 = l0_resident_grid(bn_bwd_reduce_l0_kernel<uint16_t>, h, SS.n, (long)N * H, W);
      const int nb2 = l0_resident_grid(bn_bwd_apply_l0_kernel<uint16_t>, h, SS.n, (long)N * H, W);
      if (!sums_fused)
        hipLaunchKernelGGL(bn_bwd_reduce_l0_kernel<uint16_t>, dim3(nb1, SS.n), dim3(256), l0_lds_bytes(W), st, a0, a1, P(h, d.w_off), P(h, d.b_off));
      hipLaunchKernelGGL(bn_bwd_sums_kernel<float>, dim3(cdiv(64 * 32, 256)), dim3(256), 0, st, a0, a1, SS.n, dg, db);
      hipLaunchKernelGGL(bn_bwd_apply_l0_kernel<uint16_t>, dim3(nb2, SS.n), dim3(256), l0_lds_bytes(W), st, a0, a1, P(h, d.w_off), P(h, d.b_off),
                         Gd(h, d.w_off));
      HIPCHK(hipGetLastError());
      continue;
    }
    // pass 2 (APPLY) rides the layer's weight gradient (wgrad_bf16_kernel<.., FUSE>; SSP_BF16_FUSE_APPLY=0: the separate pass)
    const bool fuse_apply = h->bf16_fuse_apply && C % 8 == 0 && (!pool_after || ((lh | lw) & 1) == 0);
    if (pool_after) {
      // pass 1 from the raw pooled copy: the arg-max of z over a window IS that element (max for gamma >= 0, min for gamma < 0), so
      // the ReLU-layer sums over the quarter-size tensors (Apool, dOut) equal the window-routed sums over Y; channels with
      // gamma == 0 (xhat not recoverable from the pooled value) are repaired by bn_bwd_sums_kernel's scan over Y (pool_fix)
      BnBwdArgs r[2];
      for (int k = 0; k < SS.n; ++k) {
        r[k] = a[k];
        r[k].y = SS.s[k]->Apool[l]; r[k].H = lh / 2; r[k].W = lw / 2;
        a[k].pool_fix = 1;
      }
      const BnBwdArgs &a0 = a[0], &a1 = a[SS.n - 1];
      const long npix = (long)N * (lh / 2) * (lw / 2);
      const int rows = 256 / (C / 4);
      const int nb = (int)std::max(1L, std::min<long>(cdiv(npix, rows), 1024));
      if (!sums_fused)
        hipLaunchKernelGGL((bn_bwd_kernel<true, false, false, uint16_t>), dim3(nb, SS.n), dim3(256), 0, st, r[0], r[SS.n - 1], r[0], r[SS.n - 1]);
      hipLaunchKernelGGL(bn_bwd_sums_kernel<uint16_t>, dim3(cdiv(C * 32, 256)), dim3(256), 0, st, a0, a1, SS.n, dg, db);
      if (!fuse_apply) hipLaunchKernelGGL((bn_bwd_kernel<true, true, true, uint16_t>), dim3(nb, SS.n), dim3(256), 0, st, a0, a1, a0, a1);
      HIPCHK(hipGetLastError());
    } else CHK((launch_bn_bwd<true, false, uint16_t>(a, SS.n, dg, db, st, sums_fused, fuse_apply)));
    // weight gradient: X = (pooled) raw output of layer l - 1 under its BatchNorm + ReLU, dY = gQ
    const int src = l - 1;
    const bool pooled_in = layer_in_mode(l) == 2;
    {
      WgradBCall w;
      w.nviews = SS.n; w.N = N; w.H = lh; w.W = lw; w.ks = 3; w.in_mode = 1;
      w.x_cs = d.cin; w.x_co = 0; w.cin = d.cin; w.dy_cs = C; w.dy_co = 0; w.cout = C; w.dw = Gd(h, d.w_off);
      for (int k = 0; k < SS.n; ++k) {
        Slot& S = *SS.s[k];
        if (pooled_in && !S.pool_raw[src]) return fail(-3, "bf16 path: layer %d has no raw pooled output", src);
        w.x[k] = pooled_in ? S.Apool[src] : S.Y[src]; w.dy[k] = S.gQ; w.x_scale[k] = S.bn[src].scale; w.x_shift[k] = S.bn[src].shift;
      }
      if (src >= 5 && S0.act_valid[src] && SS.s[SS.n - 1]->act_valid[src]) {   // the forward materialised this layer's activated input
        w.in_mode = 0;
        for (int k = 0; k < SS.n; ++k) w.x[k] = SS.s[k]->act[src];
      }
      if (fuse_apply) {
        w.fuse = pool_after ? 2 : 1; w.f_gamma = P(h, d.g_off); w.f_dcs = C; w.f_dco = 0;
        for (int k = 0; k < SS.n; ++k) {
          Slot& S = *SS.s[k];
          w.f_y[k] = S.Y[l]; w.f_dout[k] = S.gP; w.f_scale[k] = S.bn[l].scale; w.f_shift[k] = S.bn[l].shift; w.f_mean[k] = S.bn[l].mean;
          w.f_invstd[k] = S.bn[l].invstd; w.f_k12[k] = S.bn[l].k12;
        }
      }
      const double flops = 2.0 * SS.n * N * lh * lw * (double)d.cin * C * 9;
      // (bytes: X + dY once; with the fused APPLY: y + the gradient wrt the (pooled) activation in, dY out - counted since round 6)
      const double wbytes = 2.0 * SS.n * N * lh * lw * ((double)d.cin + (fuse_apply ? (2.0 + (pool_after ? 0.25 : 1.0)) * C : (double)C));
      ProfScope ps(h, SSP_PROF_CONV3X3_WGRAD, st, flops, wbytes, flops, SSP_PROF_K_WGRAD_BF16);
      CHK(launch_wgrad_bf16(w, h->partial, h->partial_floats, h->n_cu, st, h->rq_bf16));
    }
    {
      ConvBCall c;
      c.nviews = SS.n; c.N = N; c.H = lh; c.W = lw; c.ks = 3; c.in_mode = 0;
      c.in_cs = C; c.in_co = 0; c.cin = C; c.wpk = reinterpret_cast<const uint16_t*>(h->wpk_bwd + d.pk_bwd);
      c.out_cs = d.cin; c.out_co = 0; c.cout = d.cin;
      for (int k = 0; k < SS.n; ++k) { c.in[k] = SS.s[k]->gQ; c.out[k] = SS.s[k]->gP; }
      // pass 1 of layer src's BatchNorm backward in this launch's copy-out: its output IS dOut of layer src, at the resolution of
      // src's (pooled) raw output - the tensor the separate pass would read beside it (Apool for the pooled layers, see above)
      static const int bnr_env = getenv("SSP_BF16_BNR") ? atoi(getenv("SSP_BF16_BNR")) : 1;   // (perf-debug: 0 = separate pass 1)
      const bool bnr = bnr_env != 0 && launch_conv_bf16_is_ws(c);
      if (bnr) {
        for (int k = 0; k < SS.n; ++k) {
          Slot& S = *SS.s[k];
          c.bnr_t[k] = reinterpret_cast<const uint16_t*>(pooled_in ? S.Apool[src] : S.Y[src]);
          c.bnr_scale[k] = S.bn[src].scale; c.bnr_shift[k] = S.bn[src].shift; c.bnr_mean[k] = S.bn[src].mean;
          c.bnr_invstd[k] = S.bn[src].invstd; c.bnr_sums[k] = S.bn[src].bsums;
        }
      }
      const double flops = 2.0 * SS.n * N * lh * lw * (double)d.cin * C * 9;
      // (bytes: dY in, dOut of the layer below out, + the layer-below raw (pooled) output the fused BatchNorm-backward sums read)
      ProfScope ps(h, SSP_PROF_CONV3X3_DGRAD, st, flops, 2.0 * SS.n * N * lh * lw * ((double)d.cin * (bnr ? 2.0 : 1.0) + C), flops, SSP_PROF_K_CONV_BF16);
      CHK(launch_conv_bf16(c, h->n_cu, st));
      if (bnr) h->bsums_fused[src] = true;
    }
  }
  return 0;
}

// bf16 path: backward of the heads.  fp32 BatchNorm backward of Pb / Db (their outputs feed the fp32 losses), pointwise weight and
// data gradients with the fp32 dY rounded to bf16 on load (-> gP = bf16 gradient wrt the activations of the 3x3 heads,
// [cells][256 heads]), BatchNorm + ReLU backward of Pa / Da / DS on the bf16 tensors (-> gQ), their weight gradients and ONE data
// gradient over the concatenated dY channels (-> gP = bf16 [cells][128], gradient wrt the activation of encoder layer 7).
static int heads_backward_bf16(ssp_handle* h, const SlotSet& SS, const float* const* dsemi, const float* const* draw_desc,
                               float* const* dsout, hipStream_t st) {
  Slot& S0 = *SS.s[0];
  const int N = S0.N, Hc = S0.H / 8, Wc = S0.W / 8, hcs = 256 * h->nheads;
  const size_t ncells = (size_t)N * Hc * Wc;
  const bool has_semi = dsemi[0] != nullptr, has_desc = draw_desc[0] != nullptr, has_sem = dsout[0] != nullptr && h->nheads == 3;
  float *gQs[2] = {nullptr, nullptr}, *gQd[2] = {nullptr, nullptr};
  for (int k = 0; k < SS.n; ++k) { gQs[k] = SS.s[k]->gQ; gQd[k] = SS.s[k]->gQ + ncells * 80; }
  auto pointwise = [&](int l, int src, const float* const* dy, int dy_cs, int co) -> int {
    const LayerDesc& d = h->L[l];
    WgradBCall w;
    w.nviews = SS.n; w.N = N; w.H = Hc; w.W = Wc; w.ks = 1; w.in_mode = 1; w.dy_f32 = true;
    w.x_cs = hcs; w.x_co = co; w.cin = 256; w.dy_cs = dy_cs; w.dy_co = 0; w.cout = d.cout; w.dw = Gd(h, d.w_off);
    ConvBCall c;
    c.nviews = SS.n; c.N = N; c.H = Hc; c.W = Wc; c.ks = 1; c.in_mode = 0; c.in_f32 = true;
    c.in_cs = dy_cs; c.in_co = 0; c.cin = d.cout; c.wpk = reinterpret_cast<const uint16_t*>(h->wpk_bwd + d.pk_bwd);
    c.out_cs = hcs; c.out_co = co; c.cout = 256;
    // pass 1 of the 3x3 head's BatchNorm backward in this data gradient's copy-out (its output is dOut of that head; the head's raw
    // output has the output's geometry: its 256-channel slice of [cells][256 heads])
    static const int bnr_env = getenv("SSP_BF16_BNR") ? atoi(getenv("SSP_BF16_BNR")) : 1;
    // (the same geometry predicate as the fp32 path's add_dgrad: the generic kernel reads S.Y[src] with the OUTPUT's stride / offset /
    // channel count - a layer table in which the head's raw output sits elsewhere falls back to the separate pass 1)
    const bool bnr = bnr_env != 0 && h->L[src].bn && S0.y_cs[src] == hcs && S0.y_co[src] == co && h->L[src].cout == c.cout;
    for (int k = 0; k < SS.n; ++k) {
      Slot& S = *SS.s[k];
      w.x[k] = S.Y[src]; w.dy[k] = dy[k]; w.x_scale[k] = S.bn[src].scale; w.x_shift[k] = S.bn[src].shift;
      c.in[k] = dy[k]; c.out[k] = S.gP;
      if (bnr) {
        c.bnr_t[k] = reinterpret_cast<const uint16_t*>(S.Y[src]);
        c.bnr_scale[k] = S.bn[src].scale; c.bnr_shift[k] = S.bn[src].shift; c.bnr_mean[k] = S.bn[src].mean;
        c.bnr_invstd[k] = S.bn[src].invstd; c.bnr_sums[k] = S.bn[src].bsums;
      }
    }
    CHK(launch_wgrad_bf16(w, h->partial, h->partial_floats, h->n_cu, st, h->rq_bf16));
    CHK(launch_conv_bf16(c, h->n_cu, st));
    if (bnr) h->bsums_fused[src] = true;
    return 0;
  };
  // (the two fp32 BatchNorm backward passes of the pointwise heads in three launches for both, like the fp32 path)
  BnPlainJob job_pb, job_db;
  const bool pair = has_semi && has_desc && SS.n == 2;
  if (pair) {
    CHK(bn_layer_backward(h, SS, L_PB, dsemi, 80, 0, false, false, gQs, 80, 0, N, Hc, Wc, st, nullptr, &job_pb));
    CHK(bn_layer_backward(h, SS, L_DB, draw_desc, 256, 0, false, false, gQd, 256, 0, N, Hc, Wc, st, nullptr, &job_db));
    if (!job_pb.filled || !job_db.filled) return fail(-3, "pointwise heads: BatchNorm backward not in the plain form");
    CHK(launch_bn_bwd_pair(job_pb, job_db, SS.n, st));
  }
  if (has_semi) {
    if (!pair) CHK(bn_layer_backward(h, SS, L_PB, dsemi, 80, 0, false, false, gQs, 80, 0, N, Hc, Wc, st));
    CHK(pointwise(L_PB, L_PA, gQs, 80, 0));
  }
  if (has_desc) {
    if (!pair) CHK(bn_layer_backward(h, SS, L_DB, draw_desc, 256, 0, false, false, gQd, 256, 0, N, Hc, Wc, st));
    CHK(pointwise(L_DB, L_DA, gQd, 256, 256));
  }
  if (has_sem) {
    const LayerDesc& d = h->L[L_SOUT];
    if (h->sout_cs > 256) return fail(-3, "segmentation head: more than 256 classes are not supported by colsum_kernel");
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv((long)ncells, COLSUM_ROWS), SS.n), dim3(256), 0, st, dsout[0], Gd(h, d.b_off), (int)ncells, d.cout,
                       h->sout_cs, dsout[SS.n - 1]);
    HIPCHK(hipGetLastError());
    CHK(pointwise(L_SOUT, L_DS, dsout, h->sout_cs, 512));
  }
  // ---- 3x3 heads ----
  const int heads[3] = {L_PA, L_DA, L_DS};
  // pass 2 (APPLY) of their BatchNorm + ReLU backward rides their weight gradients (as in encoder_backward_bf16): y, dOut and dY are
  // the head's 256-channel slice of [cells][256 heads] tensors
  const bool heads_fuse = h->bf16_fuse_apply;
  BnSumsQueue sums_queue;   // the heads' replica reductions in one launch where nothing else is left of their BatchNorm backward
  for (int hk = 0; hk < h->nheads; ++hk) {
    const LayerDesc& d = h->L[heads[hk]];
    BnBwdArgs a[2];
    for (int k = 0; k < SS.n; ++k) {
      Slot& S = *SS.s[k];
      BnBwdArgs& v = a[k];
      const int l = heads[hk];
      v.y = S.Y[l]; v.dout = S.gP; v.dy = S.gQ; v.scale = S.bn[l].scale; v.shift = S.bn[l].shift; v.mean = S.bn[l].mean;
      v.invstd = S.bn[l].invstd; v.gamma = P(h, d.g_off); v.sums = S.bn[l].bsums; v.dbias = Gd(h, d.b_off);
      v.x = nullptr; v.apool = nullptr; v.beta = P(h, d.be_off); v.pool_fix = 0;
      v.N = N; v.H = Hc; v.W = Wc; v.C = 256; v.y_cs = hcs; v.y_co = 256 * hk; v.d_cs = hcs; v.d_co = 256 * hk;
      v.dy_cs = hcs; v.dy_co = 256 * hk; v.count = (double)ncells; v.k12 = S.bn[l].k12;
    }
    const bool sums_fused = h->bsums_fused[heads[hk]];
    h->bsums_fused[heads[hk]] = false;
    CHK((launch_bn_bwd<true, false, uint16_t>(a, SS.n, Gd(h, d.g_off), Gd(h, d.be_off), st, sums_fused, heads_fuse, &sums_queue)));
  }
  CHK(flush_bn_sums(sums_queue, SS.n, st));
  for (int hk = 0; hk < h->nheads; ++hk) {
    const LayerDesc& d = h->L[heads[hk]];
    WgradBCall w;
    w.nviews = SS.n; w.N = N; w.H = Hc; w.W = Wc; w.ks = 3; w.in_mode = 1;
    w.x_cs = 128; w.x_co = 0; w.cin = 128; w.dy_cs = hcs; w.dy_co = 256 * hk; w.cout = 256; w.dw = Gd(h, d.w_off);
    const bool from_act7 = S0.act_valid[7] && SS.s[SS.n - 1]->act_valid[7];   // the forward materialised the activated input
    if (from_act7) w.in_mode = 0;
    for (int k = 0; k < SS.n; ++k) {
      Slot& S = *SS.s[k];
      w.x[k] = from_act7 ? S.act[7] : S.Y[7]; w.dy[k] = S.gQ; w.x_scale[k] = S.bn[7].scale; w.x_shift[k] = S.bn[7].shift;
    }
    if (heads_fuse) {
      const int l = heads[hk];
      w.fuse = 1; w.f_gamma = P(h, d.g_off); w.f_dcs = hcs; w.f_dco = 256 * hk;
      for (int k = 0; k < SS.n; ++k) {
        Slot& S = *SS.s[k];
        w.f_y[k] = S.Y[l]; w.f_dout[k] = S.gP; w.f_scale[k] = S.bn[l].scale; w.f_shift[k] = S.bn[l].shift; w.f_mean[k] = S.bn[l].mean;
        w.f_invstd[k] = S.bn[l].invstd; w.f_k12[k] = S.bn[l].k12;
      }
    }
    const double flops = 2.0 * SS.n * ncells * 128.0 * 256 * 9;
    ProfScope ps(h, SSP_PROF_CONV3X3_WGRAD, st, flops, 2.0 * SS.n * ncells * (128.0 + 256.0), flops, SSP_PROF_K_WGRAD_BF16);
    CHK(launch_wgrad_bf16(w, h->partial, h->partial_floats, h->n_cu, st, h->rq_bf16));
  }
  ConvBCall c;
  c.nviews = SS.n; c.N = N; c.H = Hc; c.W = Wc; c.ks = 3; c.in_mode = 0;
  c.in_cs = hcs; c.in_co = 0; c.cin = hcs; c.wpk = reinterpret_cast<const uint16_t*>(h->wpk_heads_bwd);
  c.out_cs = 128; c.out_co = 0; c.cout = 128;
  for (int k = 0; k < SS.n; ++k) { c.in[k] = SS.s[k]->gQ; c.out[k] = SS.s[k]->gP; }
  // pass 1 of layer 7's BatchNorm backward in this launch's copy-out (its output IS dOut of layer 7), as between the encoder layers
  static const int bnr_env = getenv("SSP_BF16_BNR") ? atoi(getenv("SSP_BF16_BNR")) : 1;
  const bool bnr = bnr_env != 0 && launch_conv_bf16_is_ws(c);
  if (bnr) {
    for (int k = 0; k < SS.n; ++k) {
      Slot& S = *SS.s[k];
      c.bnr_t[k] = reinterpret_cast<const uint16_t*>(S.Y[7]);
      c.bnr_scale[k] = S.bn[7].scale; c.bnr_shift[k] = S.bn[7].shift; c.bnr_mean[k] = S.bn[7].mean;
      c.bnr_invstd[k] = S.bn[7].invstd; c.bnr_sums[k] = S.bn[7].bsums;
    }
  }
  const double flops = 2.0 * SS.n * ncells * (double)hcs * 128 * 9;
  ProfScope ps(h, SSP_PROF_CONV3X3_DGRAD, st, flops, 2.0 * SS.n * ncells * (hcs + 128.0), flops, SSP_PROF_K_CONV_BF16);
  CHK(launch_conv_bf16(c, h->n_cu, st));
  if (bnr) h->bsums_fused[7] = true;
  return 0;
}

// dsemi[k]: [cells][80] grad wrt semi (post bnPb); draw_desc[k]: [cells][256] grad wrt bnDb output (pre-normalisation);
// dsout[k]: [cells][sout_cs] grad wrt convSout output (ssmall).  A null entry means "no gradient from that head" and
// must be null for every view of the set.
// part: 0 = everything; 1 = heads + encoder layers 7..EARLY_SPLIT_LAYER, then the pending weight-gradient slabs are
// reduced, so every gradient from layer EARLY_SPLIT_LAYER's conv weight to the end of the flat vector is FINAL (the
// early all-reduce bucket, ssp_grad_early_offset); 2 = the remaining layers EARLY_SPLIT_LAYER-1..0 (their dOut sits in gP).
enum { EARLY_SPLIT_LAYER = 2 };
static int run_backward_impl(ssp_handle* h, const SlotSet& SS, const float* const* dsemi, const float* const* draw_desc,
                             float* const* dsout, hipStream_t st, int part);
static bool bf16_fuse_apply_env() {
  const char* const e = getenv("SSP_BF16_FUSE_APPLY");
  return e ? atoi(e) != 0 : true;
}
static int run_backward(ssp_handle* h, const SlotSet& SS, const float* const* dsemi, const float* const* draw_desc,
                        float* const* dsout, hipStream_t st, int part = 0) {
  if (part != 2) h->bf16_fuse_apply = bf16_fuse_apply_env();   // (one value for the heads and the encoder of a step, both phases)
  CHK(run_backward_impl(h, SS, dsemi, draw_desc, dsout, st, part));
  return det_fold(h->buf.grads_dev, st);   // (deterministic mode) bias / first-layer gradients scattered by fp32 atomics
}
static int run_backward_impl(ssp_handle* h, const SlotSet& SS, const float* const* dsemi, const float* const* draw_desc,
                             float* const* dsout, hipStream_t st, int part) {
  Slot& S0 = *SS.s[0];
  if (h->rq_bf16 != nullptr) { h->rq_bf16->J.n = 0; h->rq_bf16->used = 0; }   // (a failed pass must not leave slab reductions queued)
  if (part != 2) { h->tail.g1_blocks = 0; h->tail.cs_views = 0; }
  if (!h->packed_bwd || h->packed_algo != g_conv_algo)
    return fail(-3, "backward: the packed weight images do not belong to this pass (%s) - run the forward of the step again",
                !h->packed_bwd ? "the last forward packed no data-gradient weights" : "the conv algorithm changed since the forward");
  for (int k = 0; k < SS.n; ++k)
    if (SS.s[k]->N <= 0) return fail(-3, "backward: the slot holds no forward");
  const int N = S0.N, H = S0.H, W = S0.W, Hc = H / 8, Wc = W / 8;
  const int hcs = 256 * h->nheads;
  const bool has_semi = dsemi[0] != nullptr, has_desc = draw_desc[0] != nullptr;
  const bool has_sem = dsout[0] != nullptr && h->nheads == 3;
  float *gP[2] = {nullptr, nullptr}, *gQ[2] = {nullptr, nullptr};
  for (int k = 0; k < SS.n; ++k) { gP[k] = SS.s[k]->gP; gQ[k] = SS.s[k]->gQ; }
  auto encoder = [&](int l_hi, int l_lo) -> int {  // dOut (gP) -> dY (gQ) -> weight gradient + data gradient (gP)
    if (bf16_path()) {
      CHK(encoder_backward_bf16(h, SS, l_hi, l_lo, st));
      return flush_wgrad_reduce(h, st);   // the queued slab reductions of this part (heads included): one launch
    }
    for (int l = l_hi; l >= l_lo; --l) {
      int lh, lw; layer_res(l, H, W, lh, lw);
      const bool pool_after = (l == 1 || l == 3 || l == 5);
      const int C = h->L[l].cout;
      CHK(bn_layer_backward(h, SS, l, gP, C, 0, true, pool_after, gQ, C, 0, N, lh, lw, st));
      if (l > 0) CHK(conv_layer_backward(h, SS, l, l - 1, gQ, C, 0, gP, h->L[l].cin, 0, N, lh, lw, layer_in_mode(l), st));
    }
    return flush_wgrad_reduce(h, st);  // the pending Winograd weight-gradient slabs -> OIHW gradients, one launch
  };
  if (part == 2) return encoder(EARLY_SPLIT_LAYER - 1, 0);
  for (int l = 0; l < 16; ++l) h->bsums_fused[l] = h->apply_fused[l] = false;  // (a failed / aborted pass must not leave a flag behind)
  for (int k = 0; k < SS.n; ++k) {
    Slot& S = *SS.s[k];
    // the forward's single memset of the statistics region also cleared the backward sums; clear them again only
    // when this slot is back-propagated a second time (autograd retain_graph)
    if (S.bsums_dirty) {
      for (int l = 0; l < h->nlayers; ++l)
        CHK(dev_zero(S.bn[l].bsums, 2 * (size_t)h->L[l].cout * NREP * sizeof(double), st));
    }
    S.bsums_dirty = true;
    if (!has_semi || !has_desc || (h->nheads == 3 && !has_sem))
      CHK(dev_zero(S.gP, (size_t)N * Hc * Wc * hcs * sizeof(float), st));
  }
  if (bf16_path()) {
    CHK(heads_backward_bf16(h, SS, dsemi, draw_desc, dsout, st));
    if (part == 1) return encoder(7, EARLY_SPLIT_LAYER);
    return encoder(7, 0);
  }
  // ---- 1x1 heads: Pb, Db (BN, no ReLU) and Sout (bias only); gP = dHeadsAct [cells][hcs], gQ = dY scratch ----
  float* dact[2] = {gP[0], gP[1]};
  // The three data gradients ride ONE grouped launch (conv1x1_group_kernel) when pack_all wrote the images of every head that
  // has a gradient; dY of Pb and Db then sit side by side in gQ ([cells][80] | [cells][256]) until that launch has read them.
  const bool grouped = (!has_semi || h->pk_g1[L_PB]) && (!has_desc || h->pk_g1[L_DB]) && (!has_sem || h->pk_g1[L_SOUT]);
  const size_t ncells_all = (size_t)N * Hc * Wc;
  float* gQd[2] = {gQ[0], gQ[1]};  // Db's dY
  if (grouped && has_semi)
    for (int k = 0; k < SS.n; ++k) gQd[k] = gQ[k] + ncells_all * 80;
  G1Layer Lg[3];
  int ng = 0;
  // ... and so do the three weight gradients (wgrad1x1_group_kernel; SSP_G1_WGRAD=0, perf-debug: wgrad_mfma_kernel per layer)
  static const int g1_wgrad_env = getenv("SSP_G1_WGRAD") ? atoi(getenv("SSP_G1_WGRAD")) : 1;
  const bool gw = grouped && g1_wgrad_env != 0 && g1w_fits(256, 65, (long)ncells_all, hcs, std::max(256, h->sout_cs));
  G1WLayer Lw[3];
  int nw = 0;
  auto add_wgrad = [&](int l, float* const* dy, int dy_cs) {
    const LayerDesc& d = h->L[l];
    const int src = l - 1;
    G1WLayer& y = Lw[nw++];
    y.x_cs = S0.y_cs[src]; y.x_co = S0.y_co[src]; y.dy_cs = dy_cs; y.dy_co = 0; y.dw = Gd(h, d.w_off); y.N = d.cout;
    for (int k = 0; k < SS.n; ++k) {
      Slot& S = *SS.s[k];
      y.x[k] = S.Y[src]; y.scale[k] = S.bn[src].scale; y.shift[k] = S.bn[src].shift; y.dy[k] = dy[k];
    }
  };
  // ... and accumulates pass 1 of the BatchNorm backward of the 3x3 head below (Pa, Da, DS: dact has the geometry of their
  // raw outputs, [cells][hcs] at channel 256 k) on the way: bn_layer_backward of those layers then skips its reduction pass
  static const int g1_bnr_env = getenv("SSP_G1_BNR") ? atoi(getenv("SSP_G1_BNR")) : 1;  // (perf-debug: 0 = separate pass)
  auto add_dgrad = [&](int l, float* const* dy, int dy_cs, int co) {
    const LayerDesc& d = h->L[l];
    const int src = l - 1;
    G1Layer& y = Lg[ng++];
    y.in_cs = dy_cs; y.in_co = 0; y.out_cs = hcs; y.out_co = co; y.wpk = h->wpk_g1_bwd[l]; y.K = d.cout; y.N = d.cin;
    const bool bnr = g1_bnr_env != 0 && h->L[src].bn && S0.y_cs[src] == hcs && S0.y_co[src] == co && h->L[src].cout == d.cin;
    for (int k = 0; k < SS.n; ++k) {
      Slot& S = *SS.s[k];
      y.in[k] = dy[k]; y.out[k] = dact[k];
      if (bnr) {
        y.bnr_y[k] = S.Y[src]; y.bnr_scale[k] = S.bn[src].scale; y.bnr_shift[k] = S.bn[src].shift; y.bnr_mean[k] = S.bn[src].mean;
        y.bnr_invstd[k] = S.bn[src].invstd; y.stats[k] = S.bn[src].bsums;
      }
    }
    if (bnr) h->bsums_fused[src] = true;
  };
  // the two pointwise heads' BatchNorm backward (pass 1, replica reduction, pass 2) in three launches for both where both run and
  // the convolutions behind them are deferred to the grouped launches anyway
  BnPlainJob job_pb, job_db;
  const bool pair = has_semi && has_desc && grouped && gw && SS.n == 2;
  if (pair) {
    CHK(bn_layer_backward(h, SS, L_PB, dsemi, 80, 0, false, false, gQ, 80, 0, N, Hc, Wc, st, nullptr, &job_pb));
    CHK(bn_layer_backward(h, SS, L_DB, draw_desc, 256, 0, false, false, gQd, 256, 0, N, Hc, Wc, st, nullptr, &job_db));
    if (!job_pb.filled || !job_db.filled) return fail(-3, "pointwise heads: BatchNorm backward not in the plain form");
    CHK(launch_bn_bwd_pair(job_pb, job_db, SS.n, st));
  }
  if (has_semi) {
    if (!pair) CHK(bn_layer_backward(h, SS, L_PB, dsemi, 80, 0, false, false, gQ, 80, 0, N, Hc, Wc, st));
    CHK(conv_layer_backward(h, SS, L_PB, L_PA, gQ, 80, 0, dact, hcs, 0, N, Hc, Wc, 1, st, grouped, gw));
    if (grouped) add_dgrad(L_PB, gQ, 80, 0);
    if (gw) add_wgrad(L_PB, gQ, 80);
  }
  if (has_desc) {
    if (!pair) CHK(bn_layer_backward(h, SS, L_DB, draw_desc, 256, 0, false, false, gQd, 256, 0, N, Hc, Wc, st));
    CHK(conv_layer_backward(h, SS, L_DB, L_DA, gQd, 256, 0, dact, hcs, 256, N, Hc, Wc, 1, st, grouped, gw));
    if (grouped) add_dgrad(L_DB, gQd, 256, 256);
    if (gw) add_wgrad(L_DB, gQd, 256);
  }
  if (has_sem) {
    const LayerDesc& d = h->L[L_SOUT];
    const int ncells = N * Hc * Wc;
    if (h->sout_cs > 256) return fail(-3, "segmentation head: more than 256 classes are not supported by colsum_kernel");
    if (tail_merge_env()) {   // dsout outlives the pass: the column sums join the tail launch
      TailJobs& T = h->tail;
      T.cs_m[0] = dsout[0]; T.cs_m[1] = dsout[SS.n - 1]; T.cs_out = Gd(h, d.b_off);
      T.cs_rows = ncells; T.cs_C = d.cout; T.cs_cs = h->sout_cs; T.cs_blocks = cdiv(ncells, COLSUM_ROWS); T.cs_views = SS.n;
    } else {
      hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(ncells, COLSUM_ROWS), SS.n), dim3(256), 0, st, dsout[0], Gd(h, d.b_off), ncells, d.cout,
                         h->sout_cs, dsout[SS.n - 1]);
      HIPCHK(hipGetLastError());
    }
    CHK(conv_layer_backward(h, SS, L_SOUT, L_DS, dsout, h->sout_cs, 0, dact, hcs, 512, N, Hc, Wc, 1, st, grouped, gw));
    if (grouped) add_dgrad(L_SOUT, dsout, h->sout_cs, 512);
    if (gw) add_wgrad(L_SOUT, dsout, h->sout_cs);
  }
  if (gw && nw > 0) CHK(launch_g1_wgrad(Lw, nw, SS.n, (long)ncells_all, h->g1_partial, h->g1_partial_slabs, h->n_cu, st,
                                        tail_merge_env() ? &h->tail : nullptr));
  if (grouped && ng > 0) CHK(launch_g1(Lg, ng, SS.n, (long)ncells_all, 0, h->n_cu, st));
  // ---- 3x3 heads: BN+ReLU backward gP -> gQ [cells][hcs]; weight gradients; ONE data-gradient conv over the
  // concatenated dY channels (sums the heads' contributions) gQ -> gP [cells][128] ----
  {
    const int heads[3] = {L_PA, L_DA, L_DS};
    BnSumsQueue sums_queue;   // the heads' replica reductions in one launch where nothing else is left of their BatchNorm backward
    for (int hk = 0; hk < h->nheads; ++hk)
      CHK(bn_layer_backward(h, SS, heads[hk], gP, hcs, 256 * hk, true, false, gQ, hcs, 256 * hk, N, Hc, Wc, st, &sums_queue));
    CHK(flush_bn_sums(sums_queue, SS.n, st));
    for (int hk = 0; hk < h->nheads; ++hk) {
      const LayerDesc& d = h->L[heads[hk]];
      WgradCall w;
      w.in = S0.Y[7]; w.in_cs = 128; w.in_co = 0; w.cin = 128; w.dout = gQ[0]; w.dout_cs = hcs; w.dout_co = 256 * hk;
      w.cout = 256; w.in_scale = S0.bn[7].scale; w.in_shift = S0.bn[7].shift; w.dw = Gd(h, d.w_off);
      w.N = N; w.H = Hc; w.W = Wc; w.ks = 3; w.in_mode = 1;
      if (SS.n == 2) {
        Slot& B = *SS.s[1];
        w.nprob = 2; w.in2 = B.Y[7]; w.dout2 = gQ[1]; w.in_scale2 = B.bn[7].scale; w.in_shift2 = B.bn[7].shift;
      }
      if (h->apply_fused[heads[hk]]) {  // bn_layer_backward above left the APPLY pass to this launch
        h->apply_fused[heads[hk]] = false;
        w.fuse_apply = true; w.fuse_pool = false;
        w.dout_cs = hcs; w.dout_co = 0; w.f_gamma = P(h, d.g_off); w.f_ycs = hcs;
        for (int k = 0; k < SS.n; ++k) {
          Slot& S = *SS.s[k];
          (k ? w.dout2 : w.dout) = gP[k] + 256 * hk;  // the slices start at channel 256 hk of every pixel
          w.f_y[k] = S.Y[heads[hk]] + S.y_co[heads[hk]]; w.f_dy[k] = gQ[k] + 256 * hk;
          w.f_scale[k] = S.bn[heads[hk]].scale; w.f_shift[k] = S.bn[heads[hk]].shift; w.f_mean[k] = S.bn[heads[hk]].mean;
          w.f_invstd[k] = S.bn[heads[hk]].invstd; w.f_k12[k] = S.bn[heads[hk]].k12;
        }
        if (h->sums_lazy[heads[hk]]) {
          h->sums_lazy[heads[hk]] = false;
          w.f_lazy = 1; w.f_count = (double)N * Hc * Wc; w.f_dgamma = Gd(h, d.g_off); w.f_dbeta = Gd(h, d.be_off); w.f_dbias = Gd(h, d.b_off);
          for (int k = 0; k < SS.n; ++k) w.f_bsums[k] = SS.s[k]->bn[heads[hk]].bsums;
        }
      }
      CHK(launch_wgrad(h, w, h->partial, h->partial_floats, h->n_cu, st));
    }
    ConvCall c;
    c.in = gQ[0]; c.in_cs = hcs; c.in_co = 0; c.cin = hcs; c.wpk = h->wpk_heads_bwd; c.bias = nullptr; c.wino = wino_ok(3, hcs);
    c.allow_w4 = false;
    c.out = gP[0]; c.out_cs = 128; c.out_co = 0; c.cout = 128; c.in_scale = nullptr; c.in_shift = nullptr;
    c.stats = nullptr; c.N = N; c.H = Hc; c.W = Wc; c.ks = 3; c.in_mode = 0; c.nchunks = 16 * h->nheads; c.ncob = 2;
    c.backward = true;
    if (SS.n == 2) { c.nprob = 2; c.in2 = gQ[1]; c.out2 = gP[1]; }
    setup_bnr(h, SS, 7, false, c);
    CHK(launch_conv(h, c, st, SSP_PROF_CONV3X3_DGRAD));
  }
  // ---- encoder ----
  if (part == 1) return encoder(7, EARLY_SPLIT_LAYER);
  return encoder(7, 0);
}

extern "C" {

int ssp_forward(ssp_handle* h, int slot, const float* x_dev, int n, int height, int width, int train, float* semi_dev,
                float* desc_dev, float* sem_dev, void* stream) {
  if (!h || !h->bound) return fail(-1, "handle not bound");
  if (slot < 0 || slot > 1) return fail(-1, "slot must be 0 or 1");
  AlgoScope algo(h);
  if (n < 1 || n > h->cfg.max_batch || height > h->cfg.height || width > h->cfg.width || height % 8 || width % 8)
    return fail(-1, "forward shape [%d,1,%d,%d] exceeds the configured maximum or is not a multiple of 8", n, height, width);
  if ((size_t)height * width != (size_t)h->cfg.height * h->cfg.width && (size_t)n * height * width > (size_t)h->cfg.max_batch * h->cfg.height * h->cfg.width)
    return fail(-1, "forward shape too large");
  hipStream_t st = (hipStream_t)stream;
  {
    SlotSet SS{1, {&h->slot[slot], nullptr}};
    const float* xs[2] = {x_dev, nullptr};
    CHK(run_forward(h, SS, xs, n, height, width, train, h->buf.grads_dev != nullptr, st));
  }
  Slot& S = h->slot[slot];
  const int HW = (height / 8) * (width / 8);
  if (semi_dev) {
    const long tot = (long)n * 65 * HW;
    hipLaunchKernelGGL(nhwc_affine_to_nchw_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, st, S.Y[L_PB], S.bn[L_PB].scale,
                       S.bn[L_PB].shift, semi_dev, n, HW, 65, 80, 0);
  }
  if (desc_dev) {
    const long tot = (long)n * 256 * HW;
    hipLaunchKernelGGL(nhwc_affine_to_nchw_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, st, S.desc, (const float*)nullptr,
                       (const float*)nullptr, desc_dev, n, HW, 256, 256, 0);
  }
  if (sem_dev) {
    if (h->nheads != 3) return fail(-1, "sem output requested from a model without a segmentation head");
    const long tot = (long)n * h->cfg.n_classes * height * width;
    hipLaunchKernelGGL(sem_upsample_nchw_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, st, S.Y[L_SOUT], sem_dev, n,
                       height / 8, width / 8, height, width, h->cfg.n_classes, h->sout_cs);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int ssp_backward(ssp_handle* h, int slot, const float* dsemi_dev, const float* ddesc_dev, const float* dsem_dev,
                 void* stream) {
  if (!h || !h->bound || !h->buf.grads_dev) return fail(-1, "handle not bound with a gradient buffer");
  if (slot < 0 || slot > 1) return fail(-1, "slot must be 0 or 1");
  AlgoScope algo(h);
  hipStream_t st = (hipStream_t)stream;
  Slot& S = h->slot[slot];
  const int N = S.N, Hc = S.H / 8, Wc = S.W / 8, HW = Hc * Wc, ncells = N * HW;
  const float *ds = nullptr, *dd = nullptr, *dso = nullptr;
  if (dsemi_dev) {
    const long tot = (long)N * 65 * HW;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, st, dsemi_dev, S.dsemi, N, HW, 65, 80, 0);
    ds = S.dsemi;
  }
  if (ddesc_dev) {
    const long tot = (long)N * 256 * HW;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, st, ddesc_dev, S.ddesc, N, HW, 256, 256, 0);
    hipLaunchKernelGGL(desc_normalize_bwd_kernel, dim3(cdiv(ncells, 4)), dim3(256), 0, st, S.desc, S.inv_norm, S.ddesc, ncells);
    dd = S.ddesc;
  }
  if (dsem_dev) {
    if (h->nheads != 3) return fail(-1, "dsem given for a model without a segmentation head");
    CHK(dev_zero(S.dsout, (size_t)ncells * h->sout_cs * sizeof(float), st));
    const long tot = (long)N * h->cfg.n_classes * Hc * Wc;
    hipLaunchKernelGGL(sem_upsample_bwd_nchw_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, st, dsem_dev, S.dsout, N, Hc, Wc,
                       S.H, S.W, h->cfg.n_classes, h->sout_cs);
    dso = S.dsout;
  }
  HIPCHK(hipGetLastError());
  SlotSet SS{1, {&h->slot[slot], nullptr}};
  const float* dss[2] = {ds, nullptr};
  const float* dds[2] = {dd, nullptr};
  float* dsos[2] = {const_cast<float*>(dso), nullptr};
  return run_backward(h, SS, dss, dds, dsos, st);
}

int ssp_adam_step(ssp_handle* h, float lr, int step, void* stream) {
  if (!h || !h->bound || !h->buf.grads_dev || !h->buf.adam_m_dev || !h->buf.adam_v_dev)
    return fail(-1, "handle not bound with gradient and Adam state buffers");
  if (step < 1) return fail(-1, "Adam step index starts at 1");
  const long n = (long)h->n_params + 3;
  const float bc1 = 1.f - powf(0.9f, (float)step);
  const float bc2 = 1.f - powf(0.999f, (float)step);
  hipLaunchKernelGGL(adam_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, h->buf.params_dev,
                     h->buf.grads_dev, h->buf.adam_m_dev, h->buf.adam_v_dev, n, lr, bc1, sqrtf(bc2));
  HIPCHK(hipGetLastError());
  return 0;
}

}  // extern "C"

// phase 0: the whole step; 1: everything up to the point where the early gradient bucket is final; 2: the rest of the
// backward pass (encoder layers below EARLY_SPLIT_LAYER)
// Segmentation loss of one view.  algo 0: the step's choice - the (x, class) lane form (sem_ce_xc_kernel) when the label map is exactly
// 8x the logit map and the classes fit its 9 x 16 slots, the pixels-then-classes form (sem_ce_kernel) otherwise; SSP_SEM_XC=0 keeps
// the latter everywhere (same-box A/B).  1 / 2 force a form (2 fails on shapes it does not cover).
static int sem_xc_env() {
  static const int v = [] { const char* e = getenv("SSP_SEM_XC"); return e ? atoi(e) : 1; }();
  return v;
}
static int launch_sem_ce(int algo, bool train, const float* sout, const int64_t* labels, float* dsout, StepAccum* acc, int view, int B,
                         int Hc, int Wc, int H, int W, int C, int cs, hipStream_t st, const float* sout1 = nullptr,
                         const int64_t* labels1 = nullptr, float* dsout1 = nullptr) {
  // (sout1 / labels1 / dsout1: the other view of the pair, view + 1, in the same launch where the kernel can)
  const bool xc_ok = H == 8 * Hc && W == 8 * Wc && C <= 16 * SEMX_NB;
  if (algo == 2 && !xc_ok) return fail(-1, "sem_ce: the (x, class) form needs H = 8 Hc, W = 8 Wc and <= %d classes", 16 * SEMX_NB);
  if (C > SEM_MAX_C) return fail(-1, "sem_ce: at most %d classes", SEM_MAX_C);
  const bool xc = algo == 2 || (algo == 0 && xc_ok && sem_xc_env() != 0);
  const long ntile = (long)B * (Hc + 1) * (Wc + 1);
  const int nviews = sout1 != nullptr ? 2 : 1;
  if (xc) {
    static const int wgs_per_cu = [] { const char* e = getenv("SSP_SEM_XC_WGS"); return e ? atoi(e) : 2; }();
    const SemXcGeom geom = sem_xc_geom(B, Hc, Wc, wgs_per_cu * device_cu_count());   // two 4-wave workgroups per CU and view, all resident
    const int grid = B * geom.row_groups * geom.x_splits * nviews;
#define SSP_SEM_XC(MODE, NB) \
  hipLaunchKernelGGL((sem_ce_xc_kernel<MODE, NB>), dim3(grid), dim3(256), 0, st, sout, labels, dsout, acc, view, B, Hc, Wc, C, cs, geom, sout1, \
                     labels1, dsout1)
    if (train) { if (C <= 48) SSP_SEM_XC(3, 3); else if (C <= 96) SSP_SEM_XC(3, 6); else SSP_SEM_XC(3, 9); }
    else { if (C <= 48) SSP_SEM_XC(1, 3); else if (C <= 96) SSP_SEM_XC(1, 6); else SSP_SEM_XC(1, 9); }
#undef SSP_SEM_XC
  } else {
    const int grid = (int)std::min<long>((ntile + 3) / 4, 4096);
    for (int v = 0; v < nviews; ++v) {
      const float* so = v ? sout1 : sout; const int64_t* la = v ? labels1 : labels; float* ds = v ? dsout1 : dsout;
      if (train) hipLaunchKernelGGL((sem_ce_kernel<3>), dim3(grid), dim3(256), 0, st, so, la, ds, acc, view + v, B, Hc, Wc, H, W, C, cs);
      else hipLaunchKernelGGL((sem_ce_kernel<1>), dim3(grid), dim3(256), 0, st, so, la, ds, acc, view + v, B, Hc, Wc, H, W, C, cs);
    }
  }
  return 0;
}

static int pair_step_impl(ssp_handle* h, const ssp_pair_inputs* in, float* scalars_dev, int phase, hipStream_t st_in) {
  void* stream = st_in;
  if (!h || !h->bound) return fail(-1, "handle not bound");
  if (!in || !scalars_dev) return fail(-1, "null argument");
  if (phase < 0 || phase > 2) return fail(-1, "pair-step phase must be 0, 1 or 2");
  AlgoScope algo(h);
  // Single-view step (`data.warped_pair.enable: false`, Train_model_heatmap_all.py:207,237-262,330-332 - the branch the shipped
  // configs/magicpoint_shapes_pair.yaml takes): warped_image_dev == NULL.  One forward, detector (+ segmentation) loss of the
  // image only, loss_det_warp = loss_sem_warp = 0; the descriptor loss needs a pair (:343 asserts).
  const int nv = in && in->warped_image_dev ? 2 : 1;
  if (phase == 2) {
    if (!in->train) return 0;
    SlotSet SS2{nv, {&h->slot[0], nv == 2 ? &h->slot[1] : nullptr}};
    const float* none[2] = {nullptr, nullptr};
    float* nonef[2] = {nullptr, nullptr};
    return run_backward(h, SS2, none, none, nonef, (hipStream_t)stream, 2);
  }
  if (in->train && !h->buf.grads_dev) return fail(-1, "train step needs a gradient buffer");
  const int B = in->batch, H = h->cfg.height, W = h->cfg.width, Hc = H / 8, Wc = W / 8;
  if (B < 1 || B > h->cfg.max_batch || B > SSP_MAX_PAIRS)
    return fail(-1, "pair-step batch %d out of range (1..min(max_batch, %d))", B, SSP_MAX_PAIRS);
  const bool semantic = h->nheads == 3;
  if (semantic && (!in->semantic_dev || (nv == 2 && !in->warped_semantic_dev))) return fail(-1, "semantic labels required for the ssmall model");
  const bool use_desc = in->lambda_loss > 0.f;
  if (use_desc && nv == 1) return fail(-1, "need a pair of images: lambda_loss > 0 with warped_image_dev == NULL");
  if (!in->image_dev || !in->labels_dev || !in->valid_mask_dev || (nv == 2 && (!in->warped_labels_dev || !in->warped_valid_mask_dev)))
    return fail(-1, "pair step: image / labels / valid mask pointers of every view are required");
  const bool dense = use_desc && in->dense_loss != 0;
  if (dense && !h->dense_coef) return fail(-1, "dense descriptor loss needs a handle created with ssp_config.dense_loss = 1");
  if (use_desc && !dense && (!in->match_a_dev || !in->match_b_dev || !in->nonmatch_b_dev))
    return fail(-1, "sparse-loss indices required (call ssp_sample_indices or pass the reference's indices)");
  hipStream_t st = (hipStream_t)stream;
  const float* eta = h->buf.params_dev + h->n_params;
  const int ncells = B * Hc * Wc;
  hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(64), 0, st, h->accum, eta, in->multi_task, in->lambda_loss,
                     in->lamda_d, (int)semantic, B);
  SlotSet SS{nv, {&h->slot[0], nv == 2 ? &h->slot[1] : nullptr}};
  // The part of the loss phase that needs the LABELS only - cell masks, the count of the segmentation labels - runs on a third
  // stream beside the first-layer convolution and the weight packing (both leave the chip room; a launch beside the later,
  // CU-filling convolutions would delay their workgroups), and the scatter targets are zeroed by kernels that run anyway:
  // d(convSout) by the label count, d(desc) by the descriptor normalisation at the end of the forward pass.  ~0.15 ms of small
  // memory-bound launches off the serial part of the step.  SSP_EARLY_LABELS=0: in place, in front of the loss kernels.
  static const int early_env = getenv("SSP_EARLY_LABELS") ? atoi(getenv("SSP_EARLY_LABELS")) : 1;
  const bool early = early_env != 0;
  const float* masks[2] = {in->valid_mask_dev, in->warped_valid_mask_dev};
  const int64_t* sems[2] = {in->semantic_dev, in->warped_semantic_dev};
  const bool zero_dsout = semantic && in->train;
  auto label_kernels = [&](hipStream_t se) -> int {
    hipLaunchKernelGGL(cell_mask_kernel, dim3(std::min(cdiv(ncells, 4), 512), nv), dim3(256), 0, se, masks[0], h->slot[0].cellmask,
                       &h->accum->mask_cnt[0], B, H, W, masks[nv - 1], h->slot[nv - 1].cellmask, &h->accum->mask_cnt[nv - 1]);
    if (semantic)
      hipLaunchKernelGGL(sem_count_kernel, dim3(512, nv), dim3(256), 0, se, sems[0], sems[1], (long)B * H * W, h->cfg.n_classes, h->accum,
                         zero_dsout ? h->slot[0].dsout : (float*)nullptr, zero_dsout && nv == 2 ? h->slot[1].dsout : (float*)nullptr,
                         (long)ncells * h->sout_cs);
    HIPCHK(hipGetLastError());
    return 0;
  };
  if (early) {
    CHK(ensure_aux_stream(h));
    HIPCHK(hipEventRecord(h->ev_early_fork, st));   // (behind step_begin_kernel: the accumulators are zero, the last step is done)
    HIPCHK(hipStreamWaitEvent(h->aux2_stream, h->ev_early_fork, 0));
    CHK(label_kernels(h->aux2_stream));
    HIPCHK(hipEventRecord(h->ev_early_join, h->aux2_stream));
  }
  {
    const float* xs[2] = {in->image_dev, in->warped_image_dev};
    CHK(run_forward(h, SS, xs, B, H, W, 1, in->train != 0, st, false, early && in->train && use_desc && !dense));
  }
  if (early) HIPCHK(hipStreamWaitEvent(st, h->ev_early_join, 0));
  else CHK(label_kernels(st));
  {
    static const int idle_us = getenv("SSP_DBG_IDLE_US") ? atoi(getenv("SSP_DBG_IDLE_US")) : 0;
    if (idle_us > 0) hipLaunchKernelGGL(idle_kernel, dim3(1), dim3(64), 0, st, (unsigned long long)idle_us * 100ull);
  }
  // ---- descriptor loss on the side stream (fork) ----
  static const int loss_stream_env = getenv("SSP_LOSS_STREAM") ? atoi(getenv("SSP_LOSS_STREAM")) : 1;  // (perf-debug: 0 = one stream)
  hipStream_t sd = st;
  const bool forked = loss_stream_env != 0 && use_desc && !dense;  // (the dense loss reads the cell mask of the main stream's kernels)
  SideFork loss_fork;
  if (forked) {
    CHK(ensure_aux_stream(h));
    sd = h->aux_stream;
    CHK(loss_fork.fork(st, sd, h->ev_fork, h->ev_join));
  }
  if (dense) {  // utils/utils.py:779-893: Gram matrix + loss sums + d total / d dot, then the two backward GEMMs
    Slot &A = h->slot[0], &Bs = h->slot[1];
    const int pc = Hc * Wc;
    DenseArgs da;
    da.da = A.desc; da.db = Bs.desc; da.hn = in->homographies_dev; da.valid = Bs.cellmask;
    da.coef = in->train ? h->dense_coef : nullptr; da.acc = h->accum; da.B = B; da.Hc = Hc; da.Wc = Wc;
    da.lamda_d = in->dense_lamda_d; da.dist = in->descriptor_dist; da.multi_task = in->multi_task;
    hipLaunchKernelGGL(dense_dots_kernel, dim3(cdiv(pc, 64), cdiv(pc, 64), B), dim3(256), 0, sd, da);
    if (in->train) {
      hipLaunchKernelGGL((dense_grad_kernel<false>), dim3(4, cdiv(pc, 64), B), dim3(256), 0, sd, h->dense_coef, Bs.desc,
                         A.ddesc, pc);
      hipLaunchKernelGGL((dense_grad_kernel<true>), dim3(4, cdiv(pc, 64), B), dim3(256), 0, sd, h->dense_coef, A.desc,
                         Bs.ddesc, pc);
      hipLaunchKernelGGL(desc_normalize_bwd_kernel, dim3(cdiv(ncells, 4), 2), dim3(256), 0, sd, A.desc, A.inv_norm, A.ddesc, ncells,
                         Bs.desc, Bs.inv_norm, Bs.ddesc);
    }
    HIPCHK(hipGetLastError());
  } else if (use_desc) {
    Slot &A = h->slot[0], &Bs = h->slot[1];
    const int dflags = (in->sparse_method != 0 ? DESC_METHOD_1D : 0) | (in->sparse_dist != 0 ? DESC_EUCLIDEAN : 0);
    const bool euc = (dflags & DESC_EUCLIDEAN) != 0;
    const dim3 dgrid(desc_grid(B, h->cfg.n_match));
#define SSP_DESC(KERNEL, ...)                                                                 \
    {                                                                                         \
      if (euc) hipLaunchKernelGGL((KERNEL<true>), dgrid, dim3(256), 0, sd, __VA_ARGS__);      \
      else hipLaunchKernelGGL((KERNEL<false>), dgrid, dim3(256), 0, sd, __VA_ARGS__);         \
    }
#define SSP_DESC_MATCH(BWD, ...)                                                                        \
    {                                                                                                   \
      if (euc) hipLaunchKernelGGL((desc_match_kernel<BWD, true>), dgrid, dim3(256), 0, sd, __VA_ARGS__); \
      else hipLaunchKernelGGL((desc_match_kernel<BWD, false>), dgrid, dim3(256), 0, sd, __VA_ARGS__);   \
    }
    if (!in->train)   // (training: desc_match_kernel<true> below accumulates the loss sum as well)
      SSP_DESC_MATCH(false, A.desc, Bs.desc, in->match_a_dev, in->match_b_dev, (float*)nullptr, (float*)nullptr, h->accum, B, Hc, Wc,
                     h->cfg.n_match, dflags)
    SSP_DESC(desc_nonmatch_fwd_kernel, A.desc, Bs.desc, in->match_a_dev, in->nonmatch_b_dev, in->train ? h->dots : (float*)nullptr, h->accum,
             B, Hc, Wc, h->cfg.n_match, h->cfg.n_non, dflags)
    if (in->train) {
      if (!early) {
        CHK(dev_zero(A.ddesc, (size_t)ncells * 256 * sizeof(float), sd));
        CHK(dev_zero(Bs.ddesc, (size_t)ncells * 256 * sizeof(float), sd));
      }
      SSP_DESC_MATCH(true, A.desc, Bs.desc, in->match_a_dev, in->match_b_dev, A.ddesc, Bs.ddesc, h->accum, B, Hc, Wc, h->cfg.n_match, dflags)
      SSP_DESC(desc_nonmatch_bwd_kernel, A.desc, Bs.desc, in->match_a_dev, in->nonmatch_b_dev, h->dots, A.ddesc, Bs.ddesc, h->accum, B, Hc, Wc,
               h->cfg.n_match, h->cfg.n_non, dflags)
#undef SSP_DESC
#undef SSP_DESC_MATCH
      CHK(det_fold(A.ddesc, sd));   // (deterministic mode) the scattered gradients: fixed-point shadow -> tensor
      CHK(det_fold(Bs.ddesc, sd));
      hipLaunchKernelGGL(desc_normalize_bwd_kernel, dim3(cdiv(ncells, 4), 2), dim3(256), 0, sd, A.desc, A.inv_norm, A.ddesc, ncells,
                         Bs.desc, Bs.inv_norm, Bs.ddesc);
    }
    HIPCHK(hipGetLastError());
  }
  const float* labels[2] = {in->labels_dev, in->warped_labels_dev};
  {
    Slot &S = h->slot[0], &T = h->slot[nv - 1];   // both views in one launch (blockIdx.y)
    hipLaunchKernelGGL(detector_loss_kernel, dim3(std::min(cdiv(ncells, 4), 1024), nv), dim3(256), 0, st, S.Y[L_PB], S.bn[L_PB].scale,
                       S.bn[L_PB].shift, labels[0], S.cellmask, in->train ? S.dsemi : nullptr, h->accum, 0, B, H, W, 80,
                       T.Y[L_PB], T.bn[L_PB].scale, T.bn[L_PB].shift, labels[nv - 1], T.cellmask, in->train ? T.dsemi : nullptr);
  }
  HIPCHK(hipGetLastError());
  if (semantic) {   // loss sum and d(convSout) in one pass, both views in one launch (coef_sem: step_begin_kernel, sem_cnt: sem_count_kernel)
    Slot &S = h->slot[0], &T = h->slot[nv - 1];
    CHK(launch_sem_ce(0, in->train != 0, S.Y[L_SOUT], sems[0], in->train ? S.dsout : nullptr, h->accum, 0, B, Hc, Wc, H, W, h->cfg.n_classes,
                      h->sout_cs, st, nv == 2 ? T.Y[L_SOUT] : nullptr, nv == 2 ? sems[1] : nullptr, nv == 2 && in->train ? T.dsout : nullptr));
    if (in->train)
      for (int v = 0; v < nv; ++v) CHK(det_fold(h->slot[v].dsout, st));
    HIPCHK(hipGetLastError());
  }
  CHK(loss_fork.join());  // the scalars and the backward pass need both loss families
  hipLaunchKernelGGL(step_end_kernel, dim3(1), dim3(64), 0, st, h->accum, eta,
                     in->train ? h->buf.grads_dev + h->n_params : (float*)nullptr, scalars_dev, B, h->cfg.n_match,
                     in->multi_task, in->lambda_loss, in->lamda_d, (int)semantic, in->train, (int)dense, Hc * Wc, nv);
  HIPCHK(hipGetLastError());
  if (in->train) {
    const float* dss[2] = {h->slot[0].dsemi, nv == 2 ? h->slot[1].dsemi : nullptr};
    const float* dds[2] = {use_desc ? h->slot[0].ddesc : nullptr, use_desc ? h->slot[1].ddesc : nullptr};
    float* dsos[2] = {semantic ? h->slot[0].dsout : nullptr, semantic && nv == 2 ? h->slot[1].dsout : nullptr};
    CHK(run_backward(h, SS, dss, dds, dsos, st, phase));
  }
  return 0;
}

static int sample_indices_impl(ssp_handle* h, const float* homographies_dev, const float* hcell_dev, int batch, uint64_t seed,
                               const uint64_t* seed_dev, int32_t* match_a_dev, int32_t* match_b_dev, int32_t* nonmatch_b_dev,
                               hipStream_t st) {
  if (!h) return fail(-1, "null handle");
  if (!homographies_dev && !hcell_dev) return fail(-1, "sample_indices: homographies required");
  if (batch < 1 || batch > SSP_MAX_PAIRS) return fail(-1, "batch out of range (1..%d)", SSP_MAX_PAIRS);
  const int Hc = h->cfg.height / 8, Wc = h->cfg.width / 8;
  if (Hc * Wc > SAMPLER_MAX_CELLS) return fail(-1, "sampler supports at most %d cells", SAMPLER_MAX_CELLS);
  if (h->cfg.n_match > SAMPLER_MAX_CELLS) return fail(-1, "n_match too large for the device sampler");
  int cap = 2048;  // power of two >= cells and >= n_match (2048 for every reference configuration)
  while (cap < Hc * Wc || cap < h->cfg.n_match) cap <<= 1;
  static AttrOnce attr_once;
  if (attr_once.need()) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(sample_matches_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               SAMPLER_MAX_CELLS * 12));
  }
  hipLaunchKernelGGL(sample_matches_kernel, dim3(batch), dim3(1024), (size_t)cap * 12, st, homographies_dev, hcell_dev, seed,
                     seed_dev, match_a_dev, match_b_dev, Hc, Wc, h->cfg.n_match, cap);
  const long tot = (long)batch * h->cfg.n_match * h->cfg.n_non;
  hipLaunchKernelGGL(sample_nonmatches_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, st, seed, seed_dev, nonmatch_b_dev, tot, Hc,
                     Wc);
  HIPCHK(hipGetLastError());
  return 0;
}

__global__ void set_seed_kernel(uint64_t* dst, uint64_t v) { *dst = v; }

extern "C" {

int ssp_adam_step_scaled(ssp_handle* h, float lr, int step, float grad_scale, void* stream) {
  if (!h || !h->bound || !h->buf.grads_dev || !h->buf.adam_m_dev || !h->buf.adam_v_dev)
    return fail(-1, "handle not bound with gradient and Adam state buffers");
  if (step < 1) return fail(-1, "Adam step index starts at 1");
  const long n = (long)h->n_params + 3;
  const float bc1 = 1.f - powf(0.9f, (float)step);
  const float bc2 = 1.f - powf(0.999f, (float)step);
  hipLaunchKernelGGL(adam_scaled_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, h->buf.params_dev,
                     h->buf.grads_dev, h->buf.adam_m_dev, h->buf.adam_v_dev, n, lr, bc1, sqrtf(bc2), grad_scale);
  HIPCHK(hipGetLastError());
  return 0;
}

int ssp_handle_set_conv_algo(ssp_handle* h, int algo) {
  if (!h) return fail(-1, "null handle");
  if (algo < 0 || algo > 12 || algo == 4) return fail(-1, "conv algo must be 0..3 or 5..12 (see ssp_set_conv_algo)");
  if (!SSP_LEGACY_ALGOS && (algo == 2 || algo == 3 || algo == 5 || algo == 7 || algo == 8))
    return fail(-1, "conv algo %d is compiled out of the shipped library (-DSSP_LEGACY_ALGOS=1)", algo);
  h->conv_algo = algo;
  return 0;
}

int ssp_pair_step(ssp_handle* h, const ssp_pair_inputs* in, float* scalars_dev, void* stream) {
  return pair_step_impl(h, in, scalars_dev, 0, (hipStream_t)stream);
}

int ssp_pair_step_phase(ssp_handle* h, const ssp_pair_inputs* in, float* scalars_dev, int phase, void* stream) {
  return pair_step_impl(h, in, scalars_dev, phase, (hipStream_t)stream);
}

size_t ssp_grad_early_offset(const ssp_handle* h) { return h ? h->L[EARLY_SPLIT_LAYER].w_off : 0; }

// Captured form of ssp_pair_step_phase (north star: "one graph per image pair"): the first call with a given input
// signature records the launches of [device index sampling (sample_indices != 0) +] the pair step into a hipGraph; later
// calls replay it.  The sampler seed lives in device memory so that it can change between replays.
int ssp_pair_step_graph(ssp_handle* h, const ssp_pair_inputs* in, float* scalars_dev, int phase, int sample_indices,
                        void* stream) {
  if (!h || !h->bound) return fail(-1, "handle not bound");
  if (!in || !scalars_dev) return fail(-1, "null argument");
  hipStream_t st = (hipStream_t)stream;
  if (st == nullptr) return fail(-1, "ssp_pair_step_graph needs a non-default stream (stream capture)");
  // key = everything a captured launch depends on except the seed (kept in device memory)
  std::vector<unsigned char> key(sizeof(ssp_pair_inputs) + sizeof(void*) + 4 * sizeof(int));
  {
    ssp_pair_inputs k;
    memcpy(&k, in, sizeof(k));  // bytewise (padding included: the caller's struct is the key)
    k.seed = 0;
    memcpy(key.data(), &k, sizeof(k));
    memcpy(key.data() + sizeof(k), &scalars_dev, sizeof(void*));
    const int extra[4] = {phase, sample_indices, h->conv_algo, bf16_fuse_apply_env() ? 1 : 0};   // (a captured backward freezes the switch)
    memcpy(key.data() + sizeof(k) + sizeof(void*), extra, sizeof(extra));
  }
  hipGraphExec_t exec = nullptr;
  for (auto& g : h->graphs)
    if (g.key == key) { exec = g.exec; break; }
  if (exec == nullptr) {
    if (h->prof_family != 0) return fail(-1, "ssp_pair_step_graph: disable ssp_profile_enable before capturing");
    if (h->graphs.size() >= 16) {  // inputs that change every step defeat the cache: drop the oldest entry
      (void)hipGraphExecDestroy(h->graphs.front().exec);
      h->graphs.erase(h->graphs.begin());
    }
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = 0;
    if (sample_indices && phase != 2) {
      if (!in->match_a_dev || !in->match_b_dev || !in->nonmatch_b_dev) rc = fail(-1, "graph step: index buffers required");
      else rc = sample_indices_impl(h, in->homographies_dev, in->cell_homographies_dev, in->batch, 0, h->graph_seed,
                                    const_cast<int32_t*>(in->match_a_dev),
                                    const_cast<int32_t*>(in->match_b_dev), const_cast<int32_t*>(in->nonmatch_b_dev), st);
    }
    if (rc == 0) rc = pair_step_impl(h, in, scalars_dev, phase, st);
    const hipError_t ce = hipStreamEndCapture(st, &graph);
    if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (ce != hipSuccess) return fail(-2, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
    const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess) return fail(-2, "hipGraphInstantiate failed: %s", hipGetErrorString(ie));
    h->graphs.push_back({key, exec});
  }
  if (sample_indices && phase != 2) {
    hipLaunchKernelGGL(set_seed_kernel, dim3(1), dim3(1), 0, st, h->graph_seed, in->seed);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipGraphLaunch(exec, st));
  return 0;
}

int ssp_sample_indices(ssp_handle* h, const float* homographies_dev, int batch, uint64_t seed, int32_t* match_a_dev,
                       int32_t* match_b_dev, int32_t* nonmatch_b_dev, void* stream) {
  return sample_indices_impl(h, homographies_dev, nullptr, batch, seed, nullptr, match_a_dev, match_b_dev, nonmatch_b_dev,
                             (hipStream_t)stream);
}

int ssp_sample_indices_cell(ssp_handle* h, const float* cell_homographies_dev, int batch, uint64_t seed, int32_t* match_a_dev,
                            int32_t* match_b_dev, int32_t* nonmatch_b_dev, void* stream) {
  if (!cell_homographies_dev) return fail(-1, "sample_indices_cell: cell homographies required");
  return sample_indices_impl(h, nullptr, cell_homographies_dev, batch, seed, nullptr, match_a_dev, match_b_dev, nonmatch_b_dev,
                             (hipStream_t)stream);
}

// ---- operator-level entry points ------------------------------------------------------------------
int ssp_op_conv(const float* in_dev, const float* w_oihw_dev, const float* bias_dev, float* out_dev, int n, int hh, int w,
                int cin, int cout, int ksize, int in_mode, const float* in_scale_dev, const float* in_shift_dev,
                double* stats_dev, int transpose_flip, void* workspace_dev, size_t workspace_bytes, void* stream) {
  // with transpose_flip the weight tensor is [cin_conv... see header]: w is OIHW with O = (tf ? cin : cout)
  AlgoScope algo(nullptr);
  if (ksize == 1 && in_mode != 2 && g1_enabled() && g1_fits(cin, cout, (long)n * hh * w, cin, cout, 1)) {
    // pointwise layer: the grouped kernel of the heads, one problem set (conv1x1_group.hip.h); workspace = operand image
    const size_t img = g1_image_floats(cin, cout) * sizeof(float);
    if (workspace_bytes >= img) {
      hipStream_t st = (hipStream_t)stream;
      float* wpk = reinterpret_cast<float*>(workspace_dev);
      G1PackJobs G;
      G.n = 0;
      int nb = 0;
      g1_add_pack(G, nb, w_oihw_dev, wpk, transpose_flip ? cin : cout, transpose_flip ? cout : cin, transpose_flip ? 1 : 0);
      hipLaunchKernelGGL(pack_g1_kernel, dim3(nb), dim3(256), 0, st, G);
      HIPCHK(hipGetLastError());
      G1Layer y;
      y.in[0] = in_dev; y.in_cs = cin; y.in_co = 0; y.out[0] = out_dev; y.out_cs = cout; y.out_co = 0; y.wpk = wpk; y.bias = bias_dev;
      y.scale[0] = in_scale_dev; y.shift[0] = in_shift_dev; y.stats[0] = stats_dev; y.K = cin; y.N = cout;
      return launch_g1(&y, 1, 1, (long)n * hh * w, in_mode, device_cu_count(), st);
    }
  }
  const bool wino = wino_ok(ksize, cin) && in_mode != 2;
  const int nchunks = cdiv(cin, CK), ncob = cdiv(cout, NB);
  const size_t need = (size_t)ncob * nchunks * pk_taps(ksize) * CK * NB * sizeof(float);
  if (workspace_bytes < need) return fail(-4, "ssp_op_conv workspace too small (%zu < %zu)", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  float* wpk = reinterpret_cast<float*>(workspace_dev);
  const bool w4 = wino && ksize == 3 && w4_eligible(nullptr, 1, n, hh, w, cin, cout);
  if (!transpose_flip) CHK(launch_pack(w_oihw_dev, wpk, cout, cin, ksize, 0, wino, st, w4));
  else CHK(launch_pack(w_oihw_dev, wpk, cin, cout, ksize, 1, wino, st, w4));
  ConvCall c;
  c.in = in_dev; c.in_cs = cin; c.in_co = 0; c.cin = cin; c.wpk = wpk; c.bias = bias_dev; c.out = out_dev; c.out_cs = cout;
  c.out_co = 0; c.cout = cout; c.in_scale = in_scale_dev; c.in_shift = in_shift_dev; c.stats = stats_dev; c.N = n;
  c.H = hh; c.W = w; c.ks = ksize; c.in_mode = in_mode; c.nchunks = nchunks; c.ncob = ncob; c.wino = wino;
  c.backward = transpose_flip != 0;
  return launch_conv(nullptr, c, st, 0);
}

int ssp_op_conv_wgrad(const float* in_dev, const float* dout_dev, float* dw_oihw_dev, int n, int hh, int w, int cin,
                      int cout, int ksize, int in_mode, const float* in_scale_dev, const float* in_shift_dev,
                      void* workspace_dev, size_t workspace_bytes, void* stream) {
  AlgoScope algo(nullptr);
  if (ksize == 1 && in_mode == 1 && g1_enabled() && g1w_fits(cin, cout, (long)n * hh * w, cin, cout) && cdiv(cdiv(cout, 32), G1_NT) <= G1W_MAXP &&
      workspace_bytes >= (size_t)64 * G1W_SLAB * sizeof(float)) {
    // pointwise layer with 256 input channels under BatchNorm + ReLU on load: the grouped kernel of the heads
    G1WLayer y;
    y.x[0] = in_dev; y.x_cs = cin; y.x_co = 0; y.scale[0] = in_scale_dev; y.shift[0] = in_shift_dev; y.dy[0] = dout_dev;
    y.dy_cs = cout; y.dy_co = 0; y.dw = dw_oihw_dev; y.N = cout;
    return launch_g1_wgrad(&y, 1, 1, (long)n * hh * w, reinterpret_cast<float*>(workspace_dev),
                           (int)std::min<size_t>(workspace_bytes / (G1W_SLAB * sizeof(float)), 512), device_cu_count(), (hipStream_t)stream);
  }
  WgradCall c;
  c.in = in_dev; c.in_cs = cin; c.in_co = 0; c.cin = cin; c.dout = dout_dev; c.dout_cs = cout; c.dout_co = 0; c.cout = cout;
  c.in_scale = in_scale_dev; c.in_shift = in_shift_dev; c.dw = dw_oihw_dev; c.N = n; c.H = hh; c.W = w; c.ks = ksize;
  c.in_mode = in_mode;
  int n_cu = 256;
  return launch_wgrad(nullptr, c, reinterpret_cast<float*>(workspace_dev), workspace_bytes / sizeof(float), n_cu,
                      (hipStream_t)stream);
}


// ---- pair construction (row a15) ----
int ssp_op_warp_image(const float* img_dev, const float* inv_h_dev, float* out_dev, int b, int hh, int w, int nearest,
                      void* stream) {
  const long n = (long)b * hh * w;
  hipLaunchKernelGGL(warp_image_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, img_dev, inv_h_dev,
                     out_dev, b, hh, w, nearest ? 1 : 0);
  HIPCHK(hipGetLastError());
  return 0;
}

int ssp_op_erode(const float* mask_dev, float* out_dev, int b, int hh, int w, int radius, void* stream) {
  if (radius < 0 || radius > 32) return fail(-1, "erosion radius out of range");
  const long n = (long)b * hh * w;
  hipLaunchKernelGGL(erode_ellipse_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, mask_dev, out_dev, b,
                     hh, w, radius);
  HIPCHK(hipGetLastError());
  return 0;
}

static int warp_labels_impl(const float* labels_dev, const float* h_dev, const float* hpx_dev, float* out_dev, int b, int hh,
                            int w, hipStream_t st) {
  const long n = (long)b * hh * w;
  HIPCHK(hipMemsetAsync(out_dev, 0, n * sizeof(float), st));
  hipLaunchKernelGGL(warp_labels_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, labels_dev, h_dev, hpx_dev, out_dev, b, hh, w);
  HIPCHK(hipGetLastError());
  return 0;
}
int ssp_op_warp_labels(const float* labels_dev, const float* h_dev, float* out_dev, int b, int hh, int w, void* stream) {
  return warp_labels_impl(labels_dev, h_dev, nullptr, out_dev, b, hh, w, (hipStream_t)stream);
}
int ssp_op_warp_labels_px(const float* labels_dev, const float* hpx_dev, float* out_dev, int b, int hh, int w, void* stream) {
  if (!hpx_dev) return fail(-1, "warp_labels_px: pixel-space homographies required");
  return warp_labels_impl(labels_dev, nullptr, hpx_dev, out_dev, b, hh, w, (hipStream_t)stream);
}

// ---- homography-adaptation export ----
static int export_cap(const ssp_export_params* p) {
  const int d = p->nms_dist + 1;
  return cdiv(p->height, d) * cdiv(p->width, d);  // kept points are pairwise more than nms_dist apart (Chebyshev)
}
static int export_cap2(const ssp_export_params* p) {
  int c = 1;
  while (c < export_cap(p)) c <<= 1;
  return c;
}
static int export_check(const ssp_export_params* p) {
  if (!p || p->n_views < 1 || p->height < 8 || p->width < 8 || p->height % 8 || p->width % 8)
    return fail(-1, "export: n_views >= 1 and height/width multiples of 8 are required");
  if (p->nms_dist < 0 || p->nms_dist > 64 || p->border_remove < 0 || p->top_k < 0)
    return fail(-1, "export: nms_dist in [0,64], border_remove >= 0, top_k >= 0 required");
  return 0;
}
struct ExportWs {
  float *heat, *agg, *pts;
  int32_t* count;
  PointsWork pw;
  size_t bytes;
};
static ExportWs export_carve(const ssp_export_params* p, void* base) {
  Carver c{reinterpret_cast<char*>(base), 0};
  const size_t hw = (size_t)p->height * p->width;
  ExportWs w;
  w.heat = c.take<float>(hw * p->n_views);
  w.agg = c.take<float>(hw);
  w.pw.state = c.take<uint8_t>(hw);
  w.pw.cand[0] = c.take<int32_t>(hw);
  w.pw.cand[1] = c.take<int32_t>(hw);
  w.pw.keys = c.take<uint64_t>(export_cap2(p));
  w.pw.counters = c.take<int32_t>(16);
  w.pts = c.take<float>((size_t)export_cap(p) * 5);
  w.count = c.take<int32_t>(16);
  w.bytes = align_up(c.off, 256);
  return w;
}

size_t ssp_export_workspace_bytes(const ssp_export_params* p) {
  if (export_check(p)) return 0;
  return export_carve(p, nullptr).bytes;
}

int ssp_export_max_points(const ssp_export_params* p) {
  if (export_check(p)) return -1;
  const int cap = export_cap(p);
  return p->top_k > 0 ? std::min(cap, p->top_k) : cap;
}

int ssp_op_homoadapt_views(const float* img_dev, const float* inv_h_dev, float* views_dev, float* masks_dev, int n, int hh,
                           int w, void* stream) {
  const long tot = (long)n * hh * w;
  hipLaunchKernelGGL(homoadapt_views_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, img_dev, inv_h_dev,
                     views_dev, masks_dev, n, hh, w);
  HIPCHK(hipGetLastError());
  return 0;
}

int ssp_op_flatten_detection(const float* semi_nchw_dev, const float* mask_dev, float* heat_dev, int n, int hc, int wc,
                             void* stream) {
  const int ncells = n * hc * wc;
  hipLaunchKernelGGL(flatten_detection_kernel, dim3(cdiv(ncells, 4)), dim3(256), 0, (hipStream_t)stream, semi_nchw_dev,
                     (const float*)nullptr, (const float*)nullptr, mask_dev, heat_dev, ncells, hc, wc, (long)65 * hc * wc,
                     1L, (long)hc * wc);
  HIPCHK(hipGetLastError());
  return 0;
}

int ssp_op_combine_heatmap(const float* heat_dev, const float* mask_dev, const float* unwarp_h_dev, float* out_dev, int n,
                           int hh, int w, void* stream) {
  hipLaunchKernelGGL(combine_heatmap_kernel, dim3(cdiv((long)hh * w, 64)), dim3(256), 0, (hipStream_t)stream, heat_dev,
                     mask_dev, unwarp_h_dev, out_dev, n, hh, w);
  HIPCHK(hipGetLastError());
  return 0;
}

static int heatmap_points(const float* heat, const ssp_export_params* p, const PointsWork& pw, float* pts, int32_t* count,
                          hipStream_t st) {
  const int hw = p->height * p->width;
  HIPCHK(hipMemsetAsync(pw.counters, 0, 16 * sizeof(int32_t), st));
  const int cap2 = export_cap2(p);
  hipLaunchKernelGGL(nms_init_kernel, dim3(cdiv(std::max(hw, cap2), 256)), dim3(256), 0, st, heat, p->conf_thresh, pw, hw,
                     cap2);
  if (p->nms_dist <= NMS_MAX_HALO)
    hipLaunchKernelGGL(nms_tiles_kernel, dim3(cdiv(p->width, NMS_TILE) * cdiv(p->height, NMS_TILE)), dim3(1024), 0, st,
                       heat, pw, p->height, p->width, p->nms_dist, p->border_remove, cap2, 512);
  hipLaunchKernelGGL(nms_points_kernel, dim3(1), dim3(1024), 0, st, heat, pw, p->height, p->width, p->nms_dist,
                     p->border_remove, p->top_k, p->subpixel, export_cap(p), export_cap2(p), pts, count);
  HIPCHK(hipGetLastError());
  return 0;
}

int ssp_op_heatmap_points(const float* heat_dev, const ssp_export_params* p, void* workspace_dev, float* pts_dev,
                          int32_t* count_dev, void* stream) {
  CHK(export_check(p));
  if (!heat_dev || !workspace_dev || !pts_dev || !count_dev) return fail(-1, "heatmap_points: null pointer");
  return heatmap_points(heat_dev, p, export_carve(p, workspace_dev).pw, pts_dev, count_dev, (hipStream_t)stream);
}

int ssp_op_heatmap_nms(const float* heat_dev, const ssp_export_params* p, int n_maps, void* workspace_dev,
                       const float* labels_dev, float* nms_map_dev, float* pr_dev, void* stream) {
  CHK(export_check(p));
  if (!heat_dev || !workspace_dev || n_maps < 1) return fail(-1, "heatmap_nms: bad argument");
  hipStream_t st = (hipStream_t)stream;
  ssp_export_params q = *p;
  q.top_k = 0; q.subpixel = 0;
  const ExportWs w = export_carve(&q, workspace_dev);
  const size_t hw = (size_t)p->height * p->width;
  if (nms_map_dev) HIPCHK(hipMemsetAsync(nms_map_dev, 0, hw * n_maps * sizeof(float), st));
  for (int k = 0; k < n_maps; ++k) {
    CHK(heatmap_points(heat_dev + k * hw, &q, w.pw, w.pts, w.count, st));
    hipLaunchKernelGGL(nms_map_pr_kernel, dim3(1), dim3(256), 0, st, w.pts, w.count,
                       labels_dev ? labels_dev + k * hw : nullptr, nms_map_dev ? nms_map_dev + k * hw : nullptr,
                       pr_dev ? pr_dev + 2 * k : nullptr, p->height, p->width);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

int ssp_detector_heatmap(ssp_handle* h, int slot, float* heat_dev, void* stream) {
  if (!h || !h->bound) return fail(-1, "handle not bound");
  if (slot < 0 || slot > 1 || !heat_dev) return fail(-1, "detector_heatmap: bad argument");
  Slot& S = h->slot[slot];
  if (S.N <= 0) return fail(-1, "detector_heatmap: slot %d holds no forward", slot);
  const int Hc = S.H / 8, Wc = S.W / 8, ncells = S.N * Hc * Wc;
  hipLaunchKernelGGL(flatten_detection_kernel, dim3(cdiv(ncells, 4)), dim3(256), 0, (hipStream_t)stream, S.Y[L_PB],
                     S.bn[L_PB].scale, S.bn[L_PB].shift, (const float*)nullptr, heat_dev, ncells, Hc, Wc,
                     (long)Hc * Wc * S.y_cs[L_PB], (long)S.y_cs[L_PB], 1L);
  HIPCHK(hipGetLastError());
  return 0;
}

int ssp_op_soft_argmax_points(const float* heat_dev, const float* xy_dev, float* out_dev, int n, int hh, int w,
                              void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(soft_argmax_points_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, heat_dev, xy_dev,
                     out_dev, n, hh, w);
  HIPCHK(hipGetLastError());
  return 0;
}

int ssp_export_points(ssp_handle* h, const ssp_export_params* p, int n_images, const float* const* views_dev,
                      const float* const* masks_dev, const float* const* unwarp_h_dev, void* const* workspace_dev,
                      float* const* heatmap_out_dev, float* const* pts_dev, int32_t* const* count_dev, void* stream) {
  if (!h || !h->bound) return fail(-1, "handle not bound");
  CHK(export_check(p));
  if (n_images < 1 || n_images > 2) return fail(-1, "export: 1 or 2 images per call");
  AlgoScope algo(h);
  if (p->n_views > h->cfg.max_batch || (size_t)p->n_views * p->height * p->width >
                                           (size_t)h->cfg.max_batch * h->cfg.height * h->cfg.width)
    return fail(-1, "export: %d views of %dx%d exceed the configured engine size", p->n_views, p->height, p->width);
  for (int k = 0; k < n_images; ++k)
    if (!views_dev[k] || !masks_dev[k] || !unwarp_h_dev[k] || !workspace_dev[k] || !pts_dev[k] || !count_dev[k])
      return fail(-1, "export: null pointer for image %d", k);
  hipStream_t st = (hipStream_t)stream;
  SlotSet SS{n_images, {&h->slot[0], n_images == 2 ? &h->slot[1] : nullptr}};
  const float* xs[2] = {views_dev[0], n_images == 2 ? views_dev[1] : nullptr};
  // BatchNorm in train mode over the views of ONE image (models/model_wrap.py:120 leaves net.eval() commented out);
  // only the detector head is evaluated -- the export discards the descriptors (export.py:296)
  CHK(run_forward(h, SS, xs, p->n_views, p->height, p->width, 1, false, st, true));
  const int Hc = p->height / 8, Wc = p->width / 8, ncells = p->n_views * Hc * Wc;
  for (int k = 0; k < n_images; ++k) {
    Slot& S = h->slot[k];
    ExportWs w = export_carve(p, workspace_dev[k]);
    float* agg = heatmap_out_dev && heatmap_out_dev[k] ? heatmap_out_dev[k] : w.agg;
    hipLaunchKernelGGL(flatten_detection_kernel, dim3(cdiv(ncells, 4)), dim3(256), 0, st, S.Y[L_PB], S.bn[L_PB].scale,
                       S.bn[L_PB].shift, masks_dev[k], w.heat, ncells, Hc, Wc, (long)Hc * Wc * S.y_cs[L_PB],
                       (long)S.y_cs[L_PB], 1L);
    hipLaunchKernelGGL(combine_heatmap_kernel, dim3(cdiv((long)p->height * p->width, 64)), dim3(256), 0, st, w.heat,
                       masks_dev[k], unwarp_h_dev[k], agg, p->n_views, p->height, p->width);
    HIPCHK(hipGetLastError());
    CHK(heatmap_points(agg, p, w.pw, pts_dev[k], count_dev[k], st));
  }
  return 0;
}

// ---- pair construction for real data ----
int ssp_op_sample_homographies(uint64_t seed, const ssp_homography_params* p, int b, float* h_dev, float* inv_h_dev,
                               void* stream) {
  if (!p || !h_dev || !inv_h_dev || b < 1) return fail(-1, "sample_homographies: bad argument");
  if (p->n_scales < 0 || p->n_scales > 16 || p->n_angles < 0 || p->n_angles > 63 || p->patch_ratio <= 0.f || p->patch_ratio > 1.f)
    return fail(-1, "sample_homographies: n_scales <= 16, n_angles <= 63, 0 < patch_ratio <= 1 required");
  HomographyParams q;
  q.perspective = p->perspective; q.scaling = p->scaling; q.rotation = p->rotation; q.translation = p->translation;
  q.allow_artifacts = p->allow_artifacts; q.n_scales = p->n_scales; q.n_angles = p->n_angles;
  q.scaling_amplitude = p->scaling_amplitude; q.perspective_amplitude_x = p->perspective_amplitude_x;
  q.perspective_amplitude_y = p->perspective_amplitude_y; q.patch_ratio = p->patch_ratio; q.max_angle = p->max_angle;
  q.translation_overflow = p->translation_overflow;
  hipLaunchKernelGGL(sample_homographies_kernel, dim3(cdiv(b, 64)), dim3(64), 0, (hipStream_t)stream, seed, q, h_dev,
                     inv_h_dev, b);
  HIPCHK(hipGetLastError());
  return 0;
}

static int warp_labels_full_impl(const float* labels_dev, const float* h_dev, const float* hpx_dev, float* labels_out_dev,
                                 float* res_out_dev, float* bi_out_dev, int b, int hh, int w, hipStream_t st) {
  const long n = (long)b * hh * w;
  if (labels_out_dev) HIPCHK(hipMemsetAsync(labels_out_dev, 0, n * sizeof(float), st));
  if (res_out_dev) HIPCHK(hipMemsetAsync(res_out_dev, 0, 2 * n * sizeof(float), st));
  if (bi_out_dev) HIPCHK(hipMemsetAsync(bi_out_dev, 0, n * sizeof(float), st));
  hipLaunchKernelGGL(warp_labels_full_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, labels_dev, h_dev, hpx_dev, labels_out_dev,
                     res_out_dev, bi_out_dev, b, hh, w);
  HIPCHK(hipGetLastError());
  return 0;
}
int ssp_op_warp_labels_full(const float* labels_dev, const float* h_dev, float* labels_out_dev, float* res_out_dev,
                            float* bi_out_dev, int b, int hh, int w, void* stream) {
  return warp_labels_full_impl(labels_dev, h_dev, nullptr, labels_out_dev, res_out_dev, bi_out_dev, b, hh, w, (hipStream_t)stream);
}
int ssp_op_warp_labels_full_px(const float* labels_dev, const float* hpx_dev, float* labels_out_dev, float* res_out_dev,
                               float* bi_out_dev, int b, int hh, int w, void* stream) {
  if (!hpx_dev) return fail(-1, "warp_labels_full_px: pixel-space homographies required");
  return warp_labels_full_impl(labels_dev, nullptr, hpx_dev, labels_out_dev, res_out_dev, bi_out_dev, b, hh, w, (hipStream_t)stream);
}

int ssp_op_label_quantize(const float* in_dev, float* out_dev, size_t n, void* stream) {
  if (!in_dev || !out_dev) return fail(-1, "label_quantize: null pointer");
  hipLaunchKernelGGL(label_quantize_u8_kernel, dim3(cdiv((long)n, 256)), dim3(256), 0, (hipStream_t)stream, in_dev, out_dev, (long)n);
  HIPCHK(hipGetLastError());
  return 0;
}

int ssp_op_sem_finalize(const float* sem_warped_dev, const float* valid_dev, int64_t* out_dev, size_t n, int n_classes,
                        void* stream) {
  hipLaunchKernelGGL(sem_finalize_kernel, dim3(cdiv((long)n, 256)), dim3(256), 0, (hipStream_t)stream, sem_warped_dev,
                     valid_dev, out_dev, (long)n, n_classes);
  HIPCHK(hipGetLastError());
  return 0;
}

// BatchNorm(+ReLU(+2x2 max-pool)) backward as an operator: y [N,H,W,C] raw conv output, dout = gradient wrt the
// activated (pooled when pool=1: [N,H/2,W/2,C]) output; stats4 = {scale, shift, mean, invstd} each [C].
// Outputs: dy [N,H,W,C], dgamma/dbeta/dbias [C] (accumulated), sums_dev: double[2C] scratch.
int ssp_op_bn_bwd(const float* y_dev, const float* dout_dev, const float* gamma_dev, const float* stats4_dev,
                  float* dy_dev, float* dgamma_dev, float* dbeta_dev, float* dbias_dev, double* sums_dev, int n, int hh,
                  int w, int c, int relu, int pool, void* stream) {
  return ssp_op_bn_bwd_strided(y_dev, dout_dev, gamma_dev, stats4_dev, dy_dev, dgamma_dev, dbeta_dev, dbias_dev, sums_dev, n,
                               hh, w, c, c, relu, pool, stream);
}

int ssp_op_bn_bwd_strided(const float* y_dev, const float* dout_dev, const float* gamma_dev, const float* stats4_dev,
                          float* dy_dev, float* dgamma_dev, float* dbeta_dev, float* dbias_dev, double* sums_dev, int n,
                          int hh, int w, int c, int cs, int relu, int pool, void* stream) {
  if (cs < c || (cs & 3)) return fail(-1, "ssp_op_bn_bwd_strided: the channel stride must be a multiple of 4 and >= C");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemsetAsync(sums_dev, 0, 2 * (size_t)c * NREP * sizeof(double), st));
  BnBwdArgs a;
  a.y = y_dev; a.dout = dout_dev; a.dy = dy_dev; a.scale = stats4_dev; a.shift = stats4_dev + c; a.mean = stats4_dev + 2 * c;
  a.invstd = stats4_dev + 3 * c; a.gamma = gamma_dev; a.sums = sums_dev; a.dbias = dbias_dev; a.N = n; a.H = hh; a.W = w;
  a.C = c; a.y_cs = cs; a.y_co = 0; a.d_cs = cs; a.d_co = 0; a.dy_cs = cs; a.dy_co = 0; a.count = (double)n * hh * w;
  float* k12 = nullptr;
  HIPCHK(hipMallocAsync((void**)&k12, 2 * c * sizeof(float), st));
  a.k12 = k12; a.x = nullptr; a.apool = nullptr; a.beta = nullptr; a.pool_fix = 0;
  if (relu && pool) CHK((launch_bn_bwd<true, true>(&a, 1, dgamma_dev, dbeta_dev, st)));
  else if (relu) CHK((launch_bn_bwd<true, false>(&a, 1, dgamma_dev, dbeta_dev, st)));
  else CHK((launch_bn_bwd<false, false>(&a, 1, dgamma_dev, dbeta_dev, st)));
  HIPCHK(hipFreeAsync(k12, st));
  return 0;
}

// the same operator on bf16 tensors (the bf16 path: y, dout, dy are bf16 NHWC with channel stride cs; fp32 arithmetic and parameters)
int ssp_op_bn_bwd_bf16(const void* y_dev, const void* dout_dev, const float* gamma_dev, const float* stats4_dev, void* dy_dev,
                       float* dgamma_dev, float* dbeta_dev, float* dbias_dev, double* sums_dev, int n, int hh, int w, int c, int cs,
                       int relu, int pool, void* stream) {
  if (cs < c || (cs & 3)) return fail(-1, "ssp_op_bn_bwd_bf16: the channel stride must be a multiple of 4 and >= C");
  if (!relu) return fail(-3, "ssp_op_bn_bwd_bf16: only the ReLU layers run on bf16 tensors");   // (before anything is allocated)
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipMemsetAsync(sums_dev, 0, 2 * (size_t)c * NREP * sizeof(double), st));
  BnBwdArgs a;
  a.y = reinterpret_cast<const float*>(y_dev); a.dout = reinterpret_cast<const float*>(dout_dev); a.dy = reinterpret_cast<float*>(dy_dev);
  a.scale = stats4_dev; a.shift = stats4_dev + c; a.mean = stats4_dev + 2 * c;
  a.invstd = stats4_dev + 3 * c; a.gamma = gamma_dev; a.sums = sums_dev; a.dbias = dbias_dev; a.N = n; a.H = hh; a.W = w;
  a.C = c; a.y_cs = cs; a.y_co = 0; a.d_cs = cs; a.d_co = 0; a.dy_cs = cs; a.dy_co = 0; a.count = (double)n * hh * w;
  float* k12 = nullptr;
  HIPCHK(hipMallocAsync((void**)&k12, 2 * c * sizeof(float), st));
  a.k12 = k12; a.x = nullptr; a.apool = nullptr; a.beta = nullptr; a.pool_fix = 0;
  const int rc = pool ? launch_bn_bwd<true, true, uint16_t>(&a, 1, dgamma_dev, dbeta_dev, st)
                      : launch_bn_bwd<true, false, uint16_t>(&a, 1, dgamma_dev, dbeta_dev, st);
  HIPCHK(hipFreeAsync(k12, st));   // (on the failure path too)
  return rc;
}

// perf-debug hook (tools/archive/ablate_conv.py): disable parts of conv_mfma_kernel / override its grid; 0,0 = product
int ssp_set_conv_algo(int algo) {
  if (algo < 0 || algo > 12 || algo == 4)
    return fail(-1, "conv algo must be 12 (the bf16 path: bf16 activations in HBM, direct bf16 matrix-core convolutions), 0 (direct), 1 (Winograd, pipelined), 2 (Winograd, un-pipelined), 3 (Winograd, bf16 "
                    "operands), 5 (Winograd, pipelined, weights staged through LDS), 6 (Winograd, two 4-wave workgroups per CU) "
                    ", 7 (Winograd, split-bf16 hi + lo operands), 8 (forward split-bf16, backward bf16), 9 (Winograd "
                    "F(2x2,3x3) only: algorithm 1 without F(4x4,3x3) on the large maps), 10 (F(4x4,3x3) wherever legal) or 11 (algorithm 1 "
                    "with the F(3x3,4x4) weight gradient)");
  if (!SSP_LEGACY_ALGOS && (algo == 2 || algo == 3 || algo == 5 || algo == 7 || algo == 8))
    return fail(-1, "conv algo %d is compiled out of the shipped library (-DSSP_LEGACY_ALGOS=1)", algo);
  g_default_conv_algo = algo;
  return 0;
}

// test / perf-debug hook: blocks per CU the runtime admits for a kernel family (0: conv_wino_p2_kernel<1, true>,
// 1: conv_wino_pipe_kernel<1, true, true>, 2: wgrad_wino_kernel<1, true>)
int ssp_debug_occupancy(int which) {
  int n = -1;
  hipError_t e = hipErrorInvalidValue;
  if (which == 0) {
    auto k = conv_wino_p2_kernel<1, true>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS_BYTES);
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, P2_THREADS, P2_LDS_BYTES);
  } else if (which == 1) {
    auto k = conv_wino_pipe_kernel<1, true, true>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, PIPE_LDS_BYTES);
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, WINO_THREADS, PIPE_LDS_BYTES);
  } else if (which == 2) {
    auto k = wgrad_wino_kernel<1, true>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, WgradWinoGeom<true>::LDS_BYTES);
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 512, WgradWinoGeom<true>::LDS_BYTES);
  }
  else if (which == 3) {
    auto k = conv_bf16_kernel<3, 1, false, false>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, ConvBGeom<3>::LDS_BYTES);
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, ConvBGeom<3>::LDS_BYTES);
  } else if (which == 4) {
    auto k = wgrad_bf16_kernel<3, 1, false>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, WgradBGeom<3>::LDS_BYTES);
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, WgradBGeom<3>::LDS_BYTES);
  }
  if (e != hipSuccess) return fail(-2, "occupancy query failed: %s", hipGetErrorString(e));
  return n;
}

int ssp_debug_conv_knobs(int ablate, int grid) {
  g_dbg_ablate = ablate; g_dbg_grid = grid;
  return 0;
}

// debug/test hook: device pointers of internal buffers (never used by the product path)
int ssp_debug_buffer(ssp_handle* h, int slot, const char* name, float** ptr, size_t* nfloats) {
  if (!h || !h->bound || !name || !ptr) return fail(-1, "bad argument");
  Slot& S = h->slot[slot & 1];
  const size_t cells = (size_t)h->cfg.max_batch * (h->cfg.height / 8) * (h->cfg.width / 8);
  std::string n(name);
  float* p = nullptr; size_t cnt = 0;
  if (n == "gP") { p = S.gP; cnt = (size_t)h->cfg.max_batch * h->cfg.height * h->cfg.width * 64; }
  else if (n == "gQ") { p = S.gQ; cnt = (size_t)h->cfg.max_batch * h->cfg.height * h->cfg.width * 64; }
  else if (n == "dsemi") { p = S.dsemi; cnt = cells * 80; }
  else if (n == "ddesc") { p = S.ddesc; cnt = cells * 256; }
  else if (n == "desc") { p = S.desc; cnt = cells * 256; }
  else if (n == "dsout") { p = S.dsout; cnt = cells * h->sout_cs; }
  else if (n[0] == 'Y') { int l = atoi(name + 1); if (l < 0 || l >= h->nlayers) return fail(-1, "bad layer"); p = S.Y[l]; cnt = 0; }
  else if (n[0] == 'A') { int l = atoi(name + 1); if (l < 0 || l >= 8 || !S.Apool[l]) return fail(-1, "no pooled copy of that layer"); p = S.Apool[l]; cnt = 0; }
  else if (n.rfind("scale", 0) == 0) { int l = atoi(name + 5); p = S.bn[l].scale; cnt = h->L[l].cout; }
  else if (n.rfind("shift", 0) == 0) { int l = atoi(name + 5); p = S.bn[l].shift; cnt = h->L[l].cout; }
  else if (n.rfind("mean", 0) == 0) { int l = atoi(name + 4); p = S.bn[l].mean; cnt = h->L[l].cout; }
  else if (n.rfind("invstd", 0) == 0) { int l = atoi(name + 6); p = S.bn[l].invstd; cnt = h->L[l].cout; }
  else return fail(-1, "unknown buffer %s", name);
  *ptr = p;
  if (nfloats) *nfloats = cnt;
  return 0;
}

int ssp_op_dense_loss(const float* desc_a_nhwc_dev, const float* desc_b_nhwc_dev, const float* homographies_dev,
                      const float* valid_dev, int b, int hc, int wc, float lamda_d, float descriptor_dist, int multi_task,
                      float scale, void* scratch_dev, size_t scratch_bytes, float* out3_dev, float* dda_dev,
                      float* ddb_dev, void* stream) {
  const int pc = hc * wc;
  const bool grad = dda_dev != nullptr && ddb_dev != nullptr;
  const size_t need = align_up(sizeof(StepAccum), 256) + (grad ? (size_t)b * pc * pc * sizeof(float) : 0);
  if (scratch_bytes < need) return fail(-4, "ssp_op_dense_loss scratch too small (%zu < %zu)", scratch_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  StepAccum* acc = reinterpret_cast<StepAccum*>(scratch_dev);
  float* coef = grad ? reinterpret_cast<float*>(reinterpret_cast<char*>(scratch_dev) + align_up(sizeof(StepAccum), 256)) : nullptr;
  hipLaunchKernelGGL(dense_op_prep_kernel, dim3(1), dim3(256), 0, st, acc, valid_dev, b * pc, scale);
  DenseArgs da;
  da.da = desc_a_nhwc_dev; da.db = desc_b_nhwc_dev; da.hn = homographies_dev; da.valid = valid_dev; da.coef = coef;
  da.acc = acc; da.B = b; da.Hc = hc; da.Wc = wc; da.lamda_d = lamda_d; da.dist = descriptor_dist; da.multi_task = multi_task;
  hipLaunchKernelGGL(dense_dots_kernel, dim3(cdiv(pc, 64), cdiv(pc, 64), b), dim3(256), 0, st, da);
  if (grad) {
    hipLaunchKernelGGL((dense_grad_kernel<false>), dim3(4, cdiv(pc, 64), b), dim3(256), 0, st, coef, desc_b_nhwc_dev, dda_dev, pc);
    hipLaunchKernelGGL((dense_grad_kernel<true>), dim3(4, cdiv(pc, 64), b), dim3(256), 0, st, coef, desc_a_nhwc_dev, ddb_dev, pc);
  }
  hipLaunchKernelGGL(dense_op_finish_kernel, dim3(1), dim3(1), 0, st, acc, out3_dev, b, pc);
  HIPCHK(hipGetLastError());
  return 0;
}

int ssp_op_sparse_loss(const float* desc_a_nhwc_dev, const float* desc_b_nhwc_dev, const int32_t* match_a_dev,
                       const int32_t* match_b_dev, const int32_t* nonmatch_b_dev, int b, int hc, int wc, int n_match,
                       int n_non, int method, int dist, float coef_pos, float coef_neg, float* dd_a_nhwc_dev, float* dd_b_nhwc_dev,
                       float* out2_dev, void* stream) {
  if (b < 1 || b > SSP_MAX_PAIRS) return fail(-1, "batch out of range (1..%d)", SSP_MAX_PAIRS);
  if ((dd_a_nhwc_dev == nullptr) != (dd_b_nhwc_dev == nullptr)) return fail(-1, "sparse_loss: both gradient pointers or none");
  hipStream_t st = (hipStream_t)stream;
  const bool grad = dd_a_nhwc_dev != nullptr;
  const int dflags = (method != 0 ? DESC_METHOD_1D : 0) | (dist != 0 ? DESC_EUCLIDEAN : 0);
  StepAccum* acc = nullptr;
  float* dots = nullptr;
  HIPCHK(hipMallocAsync((void**)&acc, sizeof(StepAccum), st));
  HIPCHK(hipMemsetAsync(acc, 0, sizeof(StepAccum), st));
  hipLaunchKernelGGL(sparse_op_prep_kernel, dim3(1), dim3(1), 0, st, acc, coef_pos, coef_neg);
  const size_t ncell_floats = (size_t)b * hc * wc * 256;
  const bool euc = (dflags & DESC_EUCLIDEAN) != 0;
  const dim3 dgrid(desc_grid(b, n_match));
  if (grad) {
    HIPCHK(hipMallocAsync((void**)&dots, (size_t)b * n_match * n_non * sizeof(float), st));
    CHK(dev_zero(dd_a_nhwc_dev, ncell_floats * sizeof(float), st));
    CHK(dev_zero(dd_b_nhwc_dev, ncell_floats * sizeof(float), st));
  }
#define SSP_OPD(KERNEL, ...)                                                                  \
  {                                                                                           \
    if (euc) hipLaunchKernelGGL((KERNEL<true>), dgrid, dim3(256), 0, st, __VA_ARGS__);        \
    else hipLaunchKernelGGL((KERNEL<false>), dgrid, dim3(256), 0, st, __VA_ARGS__);           \
  }
  if (grad) {
    if (euc) hipLaunchKernelGGL((desc_match_kernel<true, true>), dgrid, dim3(256), 0, st, desc_a_nhwc_dev, desc_b_nhwc_dev, match_a_dev,
                                match_b_dev, dd_a_nhwc_dev, dd_b_nhwc_dev, acc, b, hc, wc, n_match, dflags);
    else hipLaunchKernelGGL((desc_match_kernel<true, false>), dgrid, dim3(256), 0, st, desc_a_nhwc_dev, desc_b_nhwc_dev, match_a_dev,
                            match_b_dev, dd_a_nhwc_dev, dd_b_nhwc_dev, acc, b, hc, wc, n_match, dflags);
  } else {
    if (euc) hipLaunchKernelGGL((desc_match_kernel<false, true>), dgrid, dim3(256), 0, st, desc_a_nhwc_dev, desc_b_nhwc_dev, match_a_dev,
                                match_b_dev, (float*)nullptr, (float*)nullptr, acc, b, hc, wc, n_match, dflags);
    else hipLaunchKernelGGL((desc_match_kernel<false, false>), dgrid, dim3(256), 0, st, desc_a_nhwc_dev, desc_b_nhwc_dev, match_a_dev,
                            match_b_dev, (float*)nullptr, (float*)nullptr, acc, b, hc, wc, n_match, dflags);
  }
  SSP_OPD(desc_nonmatch_fwd_kernel, desc_a_nhwc_dev, desc_b_nhwc_dev, match_a_dev, nonmatch_b_dev, dots, acc, b, hc, wc, n_match, n_non, dflags)
  if (grad)
    SSP_OPD(desc_nonmatch_bwd_kernel, desc_a_nhwc_dev, desc_b_nhwc_dev, match_a_dev, nonmatch_b_dev, dots, dd_a_nhwc_dev, dd_b_nhwc_dev, acc, b,
            hc, wc, n_match, n_non, dflags)
#undef SSP_OPD
  hipLaunchKernelGGL(sparse_loss_means_kernel, dim3(1), dim3(1), 0, st, acc, out2_dev, b, n_match);
  HIPCHK(hipGetLastError());
  if (dots != nullptr) HIPCHK(hipFreeAsync(dots, st));
  HIPCHK(hipFreeAsync(acc, st));
  return 0;
}

int ssp_op_detector_loss(const float* semi_nhwc_dev, int cs, const float* labels2d_dev, const float* mask2d_dev, int b, int hh,
                         int w, void* scratch_dev, size_t scratch_bytes, float* loss_dev, float* dsemi_nhwc_dev, void* stream) {
  if (hh % 8 || w % 8 || cs < 65) return fail(-1, "detector_loss: H, W multiples of 8 and channel stride >= 65 required");
  const int ncells = b * (hh / 8) * (w / 8);
  const size_t need = align_up(sizeof(StepAccum), 256) + (size_t)(ncells + 65 * 2) * sizeof(float);
  if (scratch_bytes < need) return fail(-4, "ssp_op_detector_loss scratch too small (%zu < %zu)", scratch_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  StepAccum* acc = reinterpret_cast<StepAccum*>(scratch_dev);
  float* cellmask = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch_dev) + align_up(sizeof(StepAccum), 256));
  float* ones = cellmask + ncells;  // identity BatchNorm affine: scale 1 | shift 0
  hipLaunchKernelGGL(detector_op_prep_kernel, dim3(1), dim3(64), 0, st, acc);
  hipLaunchKernelGGL(fill_affine_identity_kernel, dim3(1), dim3(128), 0, st, ones, 65);
  hipLaunchKernelGGL(cell_mask_kernel, dim3(std::min(cdiv(ncells, 4), 512)), dim3(256), 0, st, mask2d_dev, cellmask,
                     &acc->mask_cnt[0], b, hh, w);
  hipLaunchKernelGGL(detector_loss_kernel, dim3(std::min(cdiv(ncells, 4), 1024)), dim3(256), 0, st, semi_nhwc_dev, ones,
                     ones + 65, labels2d_dev, cellmask, dsemi_nhwc_dev, acc, 0, b, hh, w, cs);
  hipLaunchKernelGGL(detector_op_finish_kernel, dim3(1), dim3(64), 0, st, acc, loss_dev);
  HIPCHK(hipGetLastError());
  return 0;
}

int ssp_op_sem_loss(const float* sout_nhwc_dev, int cs, const int64_t* labels_dev, int b, int hh, int w, int n_classes, int algo,
                    void* scratch_dev, size_t scratch_bytes, float* loss_dev, float* dsout_nhwc_dev, void* stream) {
  if (hh % 8 || w % 8 || cs < n_classes || n_classes < 1) return fail(-1, "sem_loss: H, W multiples of 8 and channel stride >= n_classes required");
  if (algo < 0 || algo > 2) return fail(-1, "sem_loss: algo 0 (the step's choice), 1 or 2");
  if (scratch_bytes < sizeof(StepAccum)) return fail(-4, "ssp_op_sem_loss scratch too small (%zu < %zu)", scratch_bytes, sizeof(StepAccum));
  hipStream_t st = (hipStream_t)stream;
  StepAccum* acc = reinterpret_cast<StepAccum*>(scratch_dev);
  const int hc = hh / 8, wc = w / 8;
  hipLaunchKernelGGL(sem_op_prep_kernel, dim3(1), dim3(64), 0, st, acc);
  hipLaunchKernelGGL(sem_count_kernel, dim3(512, 1), dim3(256), 0, st, labels_dev, labels_dev, (long)b * hh * w, n_classes, acc);
  if (dsout_nhwc_dev != nullptr) CHK(dev_zero(dsout_nhwc_dev, (size_t)b * hc * wc * cs * sizeof(float), st));
  CHK(launch_sem_ce(algo, dsout_nhwc_dev != nullptr, sout_nhwc_dev, labels_dev, dsout_nhwc_dev, acc, 0, b, hc, wc, hh, w, n_classes, cs, st));
  hipLaunchKernelGGL(sem_op_finish_kernel, dim3(1), dim3(64), 0, st, acc, loss_dev);
  HIPCHK(hipGetLastError());
  return 0;
}

int ssp_op_labels(const float* labels2d_dev, const float* mask2d_dev, float* target_dev, float* cellmask_dev, int b,
                  int hh, int w, void* stream) {
  if (hh % 8 || w % 8) return fail(-1, "H and W must be multiples of 8");
  hipStream_t st = (hipStream_t)stream;
  const int ncells = b * (hh / 8) * (w / 8);
  if (labels2d_dev && target_dev)
    hipLaunchKernelGGL(labels2dto3d_kernel, dim3(cdiv(ncells, 4)), dim3(256), 0, st, labels2d_dev, target_dev, b, hh, w);
  if (mask2d_dev && cellmask_dev) {
    double* cnt = nullptr;
    HIPCHK(hipMallocAsync((void**)&cnt, sizeof(double), st));
    HIPCHK(hipMemsetAsync(cnt, 0, sizeof(double), st));
    hipLaunchKernelGGL(cell_mask_kernel, dim3(std::min(cdiv(ncells, 4), 512)), dim3(256), 0, st, mask2d_dev, cellmask_dev, cnt, b, hh, w);
    HIPCHK(hipFreeAsync(cnt, st));
  }
  HIPCHK(hipGetLastError());
  return 0;
}

}  // extern "C"
