// Host side of libcrowdmod_hip.so: model construction (mirrors the reference UNet
// constructor, /root/reference/models/backbones/unet.py:11-122), weight packing
// into MFMA fragment order, the launch sequence of one UNet forward
// (unet.py:124-167, layers.py:55-78,12-18) and the on-device reverse loops
// (models/diffusion/ddpm.py:206-282).  Everything here is plain C++ over the HIP
// runtime; the public surface is the C ABI in include/crowdmod_hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/crowdmod_hip.h"
#include "cm_kernels.h"

namespace {

thread_local std::string g_err;

int fail(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return 1;
}

#define CM_HIP(expr)                                                                      \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

constexpr int GN_GROUPS = 8;       // layers.py:9,30,41 ; unet.py:119
constexpr float GN_EPS = 1e-5f;
constexpr int ATTN_HEADS = 4;      // layers.py:10
constexpr int TIME_ROWS = 1000;    // embeddings.py:7
constexpr int MAX_SLICES = 16;
constexpr int MAX_SLOTS = 512;  // statistics slots per sample (32-row accumulator blocks of a conv)

enum KClass { K_CONV3 = 0, K_CONV1 = 1, K_NORM = 2, K_ATTN = 3, K_ELEM = 4, K_NCLASS = 8 };

struct Param {
  std::string name;
  std::vector<int64_t> shape;
  std::vector<float> host;
  bool set = false;
  int64_t numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

// A channels-last activation [B][Z][Y][X][C] plus its per-channel statistics partials.
struct Act {
  std::string name;
  float *d = nullptr;
  float *g = nullptr;     // gradient buffer (training)
  bool gset = false;      // has received a contribution in the current backward pass
  int C = 0, Z = 0, Y = 0, X = 0;
  float *part = nullptr;  // [B][nslots][C][2] per-slot (mean, M2) of every channel
  float *cnt = nullptr;   // [B][nslots] rows behind each slot
  int nslice = 1;         // slots when the stand-alone statistics kernel fills them
  int nslots = 0;         // slots of the last producer (fused conv epilogue or stats kernel)
  bool h16 = false;       // reduced-precision plan: the tensor is stored as _Float16 (same layout and strides; plan_h16)
  int aoff = -1;          // >= 0: GroupNorm statistics of this tensor go to the accumulator rows (cm_model::astat_all, channel offset aoff)
                          //   in the inference plan -- producers add exact fixed-point sums, consumers finalise, no gn_finalize launch
  int V() const { return Z * Y * X; }
};

struct BlockDesc {
  int kind;  // 0 res, 1 down, 2 up
  std::string prefix;
  int cin, cout, attention, skip, level;
};

enum OpKind { OP_CONV, OP_STATS, OP_GNFIN, OP_ATTN, OP_ATTNBLK };

struct Op {
  OpKind kind;
  int cls;
  // conv
  cm::ConvArgs ca{};
  int MB = 1, NB = 1;
  int tuned_B = -1;
  int *d_hvtab = nullptr, *d_mtab = nullptr;  // device copies of the box coordinate tables
  Act *stat_act = nullptr;  // output tensor whose GroupNorm statistics this conv produces in its epilogue
  int pm_off = -1;          // >= 0: conv_2 of a ResnetBlock; offset of its Dropout3d mask row slice (training forward)
  const Act *in0 = nullptr, *in1 = nullptr;  // forward inputs (for the backward pass)
  std::string wname, bname;
  int temb_off = -1;        // >= 0: slice of the time-embedding projection added in the epilogue
  int gn_op = -1;           // index of the OP_GNFIN op that produced this conv's on-load normalisation
  int ref_taps = 1;         // taps of the reference weight (27 even when the forward runs the 8-tap parity form)
  int ks = 1;               // K split over workgroups (tiny-spatial layers) + combine pass
  // inference-time fusion of the block's 1x1x1 skip conv into conv_2 (see ConvArgs::s2w)
  const Act *skip0 = nullptr, *skip1 = nullptr;
  std::string skip_w, skip_b;
  float *d_s2w = nullptr, *d_bias_fused = nullptr;
  bool skip_if_fused = false;  // this op is the stand-alone skip conv that a later op absorbs
  bool wino = false;        // Winograd F(2x2,3x3) kernel (cm_conv_wino.hip): full-resolution stride-1 3x3x3 layers
  float *d_wwino = nullptr;
  float *d_wwino16 = nullptr;   // the same weights as f16 operands (reduced-precision plan, cm_model_set_precision)
  float *d_wwino_b6 = nullptr;  // fp32 plan, inference forward, two-tile layers: exact bf16 x 3 split of d_wwino (pack_wino_b6)
  long long wwino_floats = 0;   // element count of d_wwino
  float *d_wfrag16 = nullptr;   // f16 fragments of a parity-form upsample conv (reduced-precision plan)
  float *d_w1x1_16 = nullptr;   // f16 fragments of a 1x1x1 conv without statistics (pack_1x1_f16; reduced-precision plan, inference)
  long long wpar_stride16 = 0;
  bool f16d = false;        // reduced-precision plan: direct f16-operand kernel (cm_conv_f16.hip) instead of the Winograd one
  int f16d_bz = 0, f16d_by = 0, f16d_bx = 0, f16d_mbw = 0;
  bool ups = false;         // parity-form upsample conv on the stage-once kernel (cm_conv_ups.hip); source tile + row blocks per wave
  int ups_tz = 0, ups_ty = 0, ups_tx = 0, ups_mbw = 0, ups_planes = 0;
  float *d_wups16 = nullptr;    // its f16 fragments under the reduced-precision plan (pack_ups_f16), floats per parity class
  long long wups16_stride = 0;
  float *d_wups_b6 = nullptr;   // fp32 plan, inference forward: bf16 x 3 split fragments (pack_ups_b6) for the six-term products
  long long wups_b6_stride = 0;
  float *d_w16d = nullptr, *d_w16d_skip = nullptr;
  bool h2_off = false;      // refresh_h2: this layer's GroupNorm affine no longer satisfies the static bound -> six-term form
  bool dbg_h2 = false;      // cm_debug_conv_io mode 2: raw sources, but the h2 form where the plan has one (the caller bounds its operands)
  bool dbg_raw = false;     // cm_debug_conv_io: the whole-sample quarter-resolution kernel without its GroupNorm (raw sources)
  bool b6d = false;         // fp32 plan: direct six-term kernel (cm_conv_b6d.hip) instead of the six-term Winograd one (inference forward)
  int b6d_bz = 0, b6d_by = 0, b6d_bx = 0, b6d_nw = 0, b6d_mbw = 0;
  // default plan, layers whose input is GroupNorm + SiLU output (bounded): f16 two-way splits, three cross terms ("h2", cm_kernels.h:
  // cm_split2_f16) instead of bf16 three-way splits, six cross terms -- same accuracy, half the matrix instructions.  Fragments of
  // w * 2^k; h2_oscale = 2^-k.  Inference-only handles (a handle that trains keeps the six-term form: its repack kernels do not
  // maintain these fragments).
  float *d_wfin_h2 = nullptr, *d_wwino_h2 = nullptr, *d_wqr_h2 = nullptr;
  float *d_wups_h2 = nullptr;   // upsample conv (raw source): h2 with a per-sample scale from the source tensor's slot statistics
  float h2_oscale = 1.f;
  bool b6s2 = false;        // fp32 plan: the stride-2 DownSample conv on the same kernel (tile fields above)
  float *d_wb6d = nullptr, *d_wb6d_skip = nullptr;
  bool qr = false;          // whole-sample kernel of the lowest resolution (cm_conv_qr.hip), inference plan
  float *d_wqr = nullptr, *d_wqr_skip = nullptr;
  float *d_wqr_b6 = nullptr;    // exact bf16 x 3 split of d_wqr for the six-term form of conv_qr2 (pack_qr_b6)
  bool train_qr = false;        // the training forward may take conv_qr2 as well (set by train_setup once the geometry is checked)
  long long wqr_floats = 0;
  bool qr_consumer = false; // OP_GNFIN whose only consumer is a qr conv: that kernel finalises the statistics itself
  bool from_slots = false;  // OP_GNFIN (inference plan): its only consumer is a six-term Winograd conv that finalises the statistics from the
                            //   producers' slot partials itself when they are few (run_conv decides per launch; set by plan_slot_consumers)
  bool fin_skipped[4] = {false, false, false, false};   // per batch lane: this step's launch of the op was skipped on that promise
  bool atomic = false;      // OP_GNFIN (inference plan): its source tensors keep accumulator statistics and its consumers finalise them -- no launch
  bool first_k = false;     // the UNet's first conv on its dedicated kernel (cm_conv_io.hip)
  int first_cin = 4;        //   input channels it contracts per tap: 4 (C <= 4) or 8
  float *d_wfirst = nullptr;
  bool small_n = false;     // <= 8 output channels: vector-ALU kernel (cm_conv_small.hip)
  float *d_wsmall = nullptr;
  int small_nco = 4;
  bool fin = false;         // the last conv on the matrix core, taps packed into the columns (cm_conv_fin.hip); falls back to small_n
  int fin_by = 0, fin_bx = 0;
  float *d_wfin = nullptr, *d_wfin16 = nullptr, *d_wfin_src = nullptr;   // six-term / f16 fragments; device copy of the reference-layout weights
  float *d_zero_bias = nullptr;
  const Act *out_act = nullptr, *resid_act = nullptr;
  double flops_per_sample = 0;
  std::string label;
  double prof_ms = 0;
  int64_t prof_n = 0;
  int prof_B = 0;           // batch of the last launch (profile report)
  // stats
  const Act *act = nullptr;
  // gn finalize
  const Act *g0 = nullptr, *g1 = nullptr;
  const float *gamma = nullptr, *beta = nullptr;
  float *gn_out = nullptr;
  float *gn_mr = nullptr;   // [B][2][Ct] group mean / rstd per channel (allocated when training)
  std::string gname, bename;  // state_dict names of gamma / beta
  // attention
  const float *qkv = nullptr;
  float *aout = nullptr;
  int S = 0, E = 0;
  // fused attention block (inference plan): replaces the four ops flagged `in_attn_block` in front of it
  bool in_attn_block = false;
  const Act *ab_x = nullptr;      // block input (and residual)
  Act *ab_out = nullptr;          // block output (statistics producer)
  int ab_gn = -1, ab_qkv = -1, ab_outc = -1;   // indices of the ops whose device parameters it reads
  float *d_win = nullptr, *d_wout = nullptr;   // reference-layout copies of in_proj_weight / out_proj.weight
  std::string win_name, wout_name;
};

}  // namespace

struct cm_schedule {
  int T = 0;
  int device = 0;
  std::vector<float> tab[6];
  float *d_sab = nullptr, *d_s1m = nullptr;
};

struct cm_train_state;
void cm_free_train_state(cm_train_state *t);  // cm_train_host.inc (host-side struct only; device buffers live in allocs)
struct cm_model {
  cm_unet_config cfg{};
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t lane_stream[4] = {nullptr, nullptr, nullptr, nullptr};  // extra lanes of the batch interleave
  hipEvent_t ev_join[4] = {nullptr, nullptr, nullptr, nullptr}, ev_fork = nullptr, ev_half = nullptr;
  int mid_at = -1;          // run_ops records ev_half after this op (once): the second lane starts half a step late
  std::vector<Param> params;
  std::map<std::string, int> pindex;
  std::vector<BlockDesc> enc, bott, dec;
  int final_ch = 0;
  bool finalized = false;
  int precision = CM_PRECISION_F32;   // matrix-core operand type of the Winograd layers (inference plan)

  std::vector<void *> allocs;
  std::vector<std::unique_ptr<Act>> acts;
  std::map<std::string, Act *> act_by_name;
  std::vector<Op> ops;
  std::map<std::string, float *> dparam;  // raw uploaded small tensors

  // persistent device buffers
  float *x8 = nullptr;          // UNet input [B][L][H][W][8]
  Act *x8_act = nullptr;
  float *eps_cl = nullptr;      // UNet output channels-last [B][L][H][W][8]
  long long *tbuf = nullptr;    // [B] timestep per sample
  float *temb_table = nullptr;  // [1000][nproj]
  int nproj = 0;
  float *d_time[7] = {nullptr};  // device copies: table, W1, b1, W2, b2, Wd_all, bd_all
  struct cm_train_state *train = nullptr;
  float *dropmask = nullptr;    // [B][nproj] Dropout3d keep-mask/(1-p) of the current training forward
  bool train_fwd = false;
  // h2 fragments are built from the weights at load time; an optimizer step leaves them behind (the device repack kernels maintain
  // the bf16 fragments only).  Stale => the inference forward runs the six-term form; the next inference entry point re-derives them
  // from the master weights (refresh_h2, cm_train_host.inc) -- once per train -> sample transition.
  bool h2_stale = false;
  // run_ops -> run_conv / fused attention: the GroupNorm finalisation op that follows a K-split layer and can ride in
  // its second pass (cm::launch_combine_gn); `fin_done` reports that it did
  // (the hand-off itself lives in thread-local variables, tl_fin_*: batch lanes enqueue from their own host threads)
  // training step: time-embedding projections of the batch computed from the live weights
  float *train_temb = nullptr;  // [B][nproj], row b
  long long *train_iota = nullptr;
  bool use_train_temb = false;
  float *mse_partial = nullptr, *mse_loss = nullptr;
  unsigned long long *astat_all = nullptr;  // accumulator statistics [max_batch][astat_C][3] (ConvArgs::astat), sample-major
  int astat_C = 0;              //   channels of all accumulator tensors together
  struct { int b0 = -1, B = 0; } astat_clean[4];   // per batch lane: the sample range whose rows are known to be zero (the last sampler step cleared them)
  float *ks_scratch = nullptr;  // raw partial outputs of K-split convs [S][B][V][Co]
  size_t ks_scratch_floats = 0;
  float *xstate = nullptr;      // sampler state [B,C,H,W,F]
  cm::StepRow *d_steptab = nullptr;  // per-step scalars of the current loop (graph replay)
  size_t steptab_cap = 0;
  int *d_kctr = nullptr;        // device-side step counter read by the table-driven step kernels
  int *d_nonfinite = nullptr;   // result word of the sampler-output health check
  float *stage_past = nullptr, *stage_fut = nullptr, *stage_out = nullptr;  // host-variant staging
  float *stage_noise = nullptr;
  size_t stage_noise_cap = 0;
  float *stage_hist = nullptr;
  size_t stage_hist_cap = 0;

  // profiling
  bool profile = false;
  float prof_ms[K_NCLASS] = {0};
  int64_t prof_n[K_NCLASS] = {0};
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> prof_events[4];   // per batch lane (each lane's thread appends to its own)
  hipEvent_t prof_base = nullptr;                 // time origin of a profiled call (recorded on the call's stream)
  float prof_union_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // per class: length of the UNION of its launch intervals over all lanes

  int L() const { return cfg.past_len + cfg.future_len; }
  int64_t per_sample() const { return (int64_t)cfg.in_channels * cfg.rows * cfg.cols * cfg.future_len; }
};

namespace {
int refresh_h2(cm_model *m);   // cm_train_host.inc: h2 fragments from the master weights of a handle that has trained

// ------------------------------------------------------------------------------
// construction
// ------------------------------------------------------------------------------
void add_param(cm_model *m, const std::string &name, std::vector<int64_t> shape) {
  Param p;
  p.name = name;
  p.shape = std::move(shape);
  p.host.assign((size_t)p.numel(), 0.f);
  m->pindex[name] = (int)m->params.size();
  m->params.push_back(std::move(p));
}

// Block wiring and state_dict names: same loop structure as unet.py:45-115.
void build_plan(cm_model *m) {
  const cm_unet_config &c = m->cfg;
  const int base = c.base_channels, nres = c.n_levels;
  const int64_t te = base, tx = (int64_t)base * c.time_multiple;
  add_param(m, "time_embeddings.time_blocks.0.weight", {TIME_ROWS, te});
  add_param(m, "time_embeddings.time_blocks.1.weight", {tx, te});
  add_param(m, "time_embeddings.time_blocks.1.bias", {tx});
  add_param(m, "time_embeddings.time_blocks.3.weight", {tx, tx});
  add_param(m, "time_embeddings.time_blocks.3.bias", {tx});
  add_param(m, "first.weight", {base, c.in_channels, 3, 3, 3});
  add_param(m, "first.bias", {base});

  std::vector<int> stack{base};
  int cin = base, idx = 0;
  for (int level = 0; level < nres; ++level) {
    const int cout = base * c.channel_mult[level];
    for (int r = 0; r < c.num_res_blocks; ++r) {
      m->enc.push_back({0, "encoder_blocks." + std::to_string(idx++), cin, cout, c.apply_attention[level] != 0, 0, level});
      cin = cout;
      stack.push_back(cin);
    }
    if (level != nres - 1) {
      m->enc.push_back({1, "encoder_blocks." + std::to_string(idx++), cin, cin, 0, 0, level});
      stack.push_back(cin);
    }
  }
  m->bott.push_back({0, "bottleneck_blocks.0", cin, cin, 1, 0, nres - 1});
  m->bott.push_back({0, "bottleneck_blocks.1", cin, cin, 0, 0, nres - 1});
  idx = 0;
  for (int level = nres - 1; level >= 0; --level) {
    const int cout = base * c.channel_mult[level];
    for (int r = 0; r < c.num_res_blocks + 1; ++r) {
      const int skip = stack.back();
      stack.pop_back();
      m->dec.push_back({0, "decoder_blocks." + std::to_string(idx++), skip + cin, cout, c.apply_attention[level] != 0, skip, level});
      cin = cout;
    }
    if (level != 0) m->dec.push_back({2, "decoder_blocks." + std::to_string(idx++), cin, cin, 0, 0, level});
  }
  m->final_ch = cin;

  auto add_block_params = [&](const BlockDesc &b) {
    const std::string &p = b.prefix;
    if (b.kind == 0) {
      add_param(m, p + ".normalize_1.weight", {b.cin});
      add_param(m, p + ".normalize_1.bias", {b.cin});
      add_param(m, p + ".conv_1.weight", {b.cout, b.cin, 3, 3, 3});
      add_param(m, p + ".conv_1.bias", {b.cout});
      add_param(m, p + ".dense_1.weight", {b.cout, tx});
      add_param(m, p + ".dense_1.bias", {b.cout});
      add_param(m, p + ".normalize_2.weight", {b.cout});
      add_param(m, p + ".normalize_2.bias", {b.cout});
      add_param(m, p + ".conv_2.weight", {b.cout, b.cout, 3, 3, 3});
      add_param(m, p + ".conv_2.bias", {b.cout});
      if (b.cin != b.cout) {
        add_param(m, p + ".match_input.weight", {b.cout, b.cin, 1, 1, 1});
        add_param(m, p + ".match_input.bias", {b.cout});
      }
      if (b.attention) {
        add_param(m, p + ".attention.group_norm.weight", {b.cout});
        add_param(m, p + ".attention.group_norm.bias", {b.cout});
        add_param(m, p + ".attention.mhsa.in_proj_weight", {3 * (int64_t)b.cout, b.cout});
        add_param(m, p + ".attention.mhsa.in_proj_bias", {3 * (int64_t)b.cout});
        add_param(m, p + ".attention.mhsa.out_proj.weight", {b.cout, b.cout});
        add_param(m, p + ".attention.mhsa.out_proj.bias", {b.cout});
      }
    } else if (b.kind == 1) {
      add_param(m, p + ".downsample.weight", {b.cout, b.cin, 3, 3, 3});
      add_param(m, p + ".downsample.bias", {b.cout});
    } else {
      add_param(m, p + ".upsample.1.weight", {b.cout, b.cin, 3, 3, 3});
      add_param(m, p + ".upsample.1.bias", {b.cout});
    }
  };
  for (auto &b : m->enc) add_block_params(b);
  for (auto &b : m->bott) add_block_params(b);
  for (auto &b : m->dec) add_block_params(b);
  add_param(m, "final.0.weight", {m->final_ch});
  add_param(m, "final.0.bias", {m->final_ch});
  add_param(m, "final.2.weight", {c.out_channels, m->final_ch, 3, 3, 3});
  add_param(m, "final.2.bias", {c.out_channels});
}

int dev_alloc(cm_model *m, void **p, size_t bytes) {
  CM_HIP(hipMalloc(p, bytes ? bytes : 4));
  m->allocs.push_back(*p);
  return 0;
}

int upload(cm_model *m, const std::vector<float> &h, float **d) {
  if (dev_alloc(m, (void **)d, h.size() * sizeof(float))) return 1;
  CM_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

Act *new_act(cm_model *m, const std::string &name, int C, int Z, int Y, int X, bool stats, int *rc) {
  auto a = std::make_unique<Act>();
  a->name = name;
  a->C = C; a->Z = Z; a->Y = Y; a->X = X;
  const size_t B = (size_t)m->cfg.max_batch;
  if (dev_alloc(m, (void **)&a->d, B * a->V() * C * sizeof(float))) { *rc = 1; return nullptr; }
  if (stats) {
    a->nslice = std::max(1, std::min(MAX_SLICES, a->V() / 128));
    if (dev_alloc(m, (void **)&a->part, B * MAX_SLOTS * C * 2 * sizeof(float))) { *rc = 1; return nullptr; }
    if (dev_alloc(m, (void **)&a->cnt, B * MAX_SLOTS * sizeof(float))) { *rc = 1; return nullptr; }
  }
  Act *r = a.get();
  m->acts.push_back(std::move(a));
  if (!name.empty()) m->act_by_name[name] = r;
  return r;
}

const Param &P(const cm_model *m, const std::string &name) { return m->params[m->pindex.at(name)]; }

// ------------------------------------------------------------------------------
// weight packing: reference [Co][Ci][kH][kW][kL] (or [Co][Ci]) -> fragment order
//   wfrag[ntile][chunk][step = tap*K8 + j][nb][lane][jj]
//     = W[co = ntile*TN + nb*32 + (lane&31)][ci = chunk*CK + 8j + 4(lane>>5) + jj][tap]
// with internal tap (dz,dy,dx) = reference [kH=dy][kW=dx][kL=dz]; zero beyond Co / Ci.
// One wave-load of a step is 64 lanes x 16 B = 1 KiB contiguous.
// ------------------------------------------------------------------------------
// Reference conv weight [Co][Ci][kH][kW][kL] -> internal tap order [Co][Ci][t], t = (dz*3 + dy)*3 + dx
// with (dz,dy,dx) = (kL,kH,kW)  (the internal layout is [Z=frames][Y=rows][X=cols]).
std::vector<float> to_internal_taps(const float *W, int Co, int Ci, int ntaps) {
  std::vector<float> out((size_t)Co * Ci * ntaps);
  for (size_t cc = 0; cc < (size_t)Co * Ci; ++cc)
    for (int t = 0; t < ntaps; ++t) {
      const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
      const int tap_ref = (ntaps == 27) ? (dy * 3 + dx) * 3 + dz : 0;
      out[cc * ntaps + t] = W[cc * ntaps + tap_ref];
    }
  return out;
}

// nn.Upsample(x2, nearest) followed by a 3x3x3 conv (layers.py:93-94) collapses, for each
// parity p of the output voxel u = 2i + p, to a 2x2x2 conv over source voxels i + e + p - 1:
// along one axis tap d of the upsampled grid reads source floor((2i + p + d - 1)/2), i.e.
//   p = 0: d=0 -> i-1 (e=0), d=1,2 -> i (e=1);    p = 1: d=0,1 -> i (e=0), d=2 -> i+1 (e=1).
// Taps that land on the same source voxel are summed here (in double, rounded once).
// in: internal order [Co][Ci][27]; out: [8 parities][Co][Ci][8], e index = (ez*2 + ey)*2 + ex.
std::vector<float> parity_weights(const std::vector<float> &Wi, int Co, int Ci) {
  std::vector<float> out((size_t)8 * Co * Ci * 8, 0.f);
  auto emap = [](int p, int d) { return p == 0 ? (d == 0 ? 0 : 1) : (d == 2 ? 1 : 0); };
  for (int p = 0; p < 8; ++p) {
    const int pz = (p >> 2) & 1, py = (p >> 1) & 1, px = p & 1;
    for (size_t cc = 0; cc < (size_t)Co * Ci; ++cc) {
      double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int dz = 0; dz < 3; ++dz)
        for (int dy = 0; dy < 3; ++dy)
          for (int dx = 0; dx < 3; ++dx)
            acc[(emap(pz, dz) * 2 + emap(py, dy)) * 2 + emap(px, dx)] += (double)Wi[cc * 27 + (dz * 3 + dy) * 3 + dx];
      for (int e = 0; e < 8; ++e) out[((size_t)p * Co * Ci + cc) * 8 + e] = (float)acc[e];
    }
  }
  return out;
}

uint16_t f32_to_f16_bits(float f);
// IEEE binary16 bits -> float (exact)
static inline float f16_bits_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, x;
  if (exp == 0) {
    if (man == 0) { x = sign; }
    else {
      int e = -1;
      do { ++e; man <<= 1; } while (!(man & 0x400u));
      x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
    }
  } else if (exp == 31) {
    x = sign | 0x7f800000u | (man << 13);
  } else {
    x = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}

// f16 version of pack_conv_weights (same order, 4 halves per lane and step), returned as floats holding two halves each
std::vector<float> pack_conv_weights_f16(const float *W, int Co, int Ci, int ntaps, int Ci_pad, int CK, int NB) {
  const int TN = 32 * NB, ntn = (Co + TN - 1) / TN, nch = Ci_pad / CK, K8 = CK / 8, nsteps = ntaps * K8;
  std::vector<uint16_t> out((size_t)ntn * nch * nsteps * NB * 64 * 4, 0);
  size_t o = 0;
  for (int nt = 0; nt < ntn; ++nt)
    for (int ch = 0; ch < nch; ++ch)
      for (int s = 0; s < nsteps; ++s) {
        const int t = s / K8, j = s % K8;
        for (int nb = 0; nb < NB; ++nb)
          for (int lane = 0; lane < 64; ++lane)
            for (int jj = 0; jj < 4; ++jj, ++o) {
              const int co = nt * TN + nb * 32 + (lane & 31);
              const int ci = ch * CK + 8 * j + 4 * (lane >> 5) + jj;
              if (co < Co && ci < Ci) out[o] = f32_to_f16_bits(W[((size_t)co * Ci + ci) * ntaps + t]);
            }
      }
  std::vector<float> packed(out.size() / 2);
  std::memcpy(packed.data(), out.data(), out.size() * 2);
  return packed;
}

// f16 fragments of ONE parity class for the stage-once upsample kernel (cm_conv_ups.hip, F16): [32-channel column block]
// [32-channel chunk][tap 8][16-channel group m][lane][8 halves], lane = 32 hh + (co % 32), ci = chunk * 32 + 16 m + 8 hh + i.
// W: [Co][Ci][8] (parity_weights of one class).  Returned as floats holding two halves each.
std::vector<float> pack_ups_f16(const float *W, int Co, int Ci) {
  const int ncb = Co / 32, nch = Ci / 32;
  std::vector<uint16_t> out((size_t)ncb * nch * 8 * 2 * 64 * 8, 0);
  size_t o = 0;
  for (int cb = 0; cb < ncb; ++cb)
    for (int ch = 0; ch < nch; ++ch)
      for (int t = 0; t < 8; ++t)
        for (int mg = 0; mg < 2; ++mg)
          for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 8; ++i, ++o) {
              const int co = cb * 32 + (lane & 31), ci = ch * 32 + 16 * mg + 8 * (lane >> 5) + i;
              out[o] = f32_to_f16_bits(W[((size_t)co * Ci + ci) * 8 + t]);
            }
  std::vector<float> packed(out.size() / 2);
  std::memcpy(packed.data(), out.data(), out.size() * 2);
  return packed;
}

// round-to-nearest-even fp32 -> bf16 (bits)
static inline uint16_t f32_to_bf16_bits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_bits_to_f32(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
// exact three-way bf16 split of an fp32 value: w = hi + mid + lo (each rounded to nearest from the running remainder;
// 8 + 8 + 8 mantissa bits, the remainders are exact in fp32)
// h2 terms of one weight (cm_kernels.h: cm_split2_f16): f16 hi / mid of w * scale, third slot zero
static inline void f16_split2(float w, float scale, uint16_t out[3]) {
  const float v = w * scale;
  out[0] = f32_to_f16_bits(v);
  out[1] = f32_to_f16_bits(v - f16_bits_to_f32(out[0]));
  out[2] = 0;
}
static inline void bf16_split3(float w, uint16_t out[3]) {
  float rem = w;
  for (int t = 0; t < 3; ++t) {
    out[t] = f32_to_bf16_bits(rem);
    rem -= bf16_bits_to_f32(out[t]);
  }
}

// bf16 x 3 fragments of ONE parity class for the stage-once upsample kernel (cm_conv_ups.hip, PREC = 2): [32-channel column
// block][32-channel chunk][tap 8][16-channel group m][term hi / mid / lo][lane][8 bf16], lane = 32 hh + (co % 32),
// ci = chunk * 32 + 16 m + 8 hh + i.  W: [Co][Ci][8] (parity_weights of one class).  Returned as floats holding two bf16 each.
// h2_scale > 0: the h2 form -- f16 hi / mid of w * h2_scale in the first two term slots (third slot zero)
std::vector<float> pack_ups_b6(const float *W, int Co, int Ci, float h2_scale = 0.f) {
  const int ncb = Co / 32, nch = Ci / 32;
  std::vector<uint16_t> out((size_t)ncb * nch * 8 * 2 * 3 * 64 * 8, 0);
  for (int cb = 0; cb < ncb; ++cb)
    for (int ch = 0; ch < nch; ++ch)
      for (int t = 0; t < 8; ++t)
        for (int mg = 0; mg < 2; ++mg)
          for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 8; ++i) {
              const int co = cb * 32 + (lane & 31), ci = ch * 32 + 16 * mg + 8 * (lane >> 5) + i;
              uint16_t t3[3];
              if (h2_scale > 0.f) f16_split2(W[((size_t)co * Ci + ci) * 8 + t], h2_scale, t3);
              else bf16_split3(W[((size_t)co * Ci + ci) * 8 + t], t3);
              for (int tm = 0; tm < 3; ++tm)
                out[(((((((size_t)cb * nch + ch) * 8 + t) * 2 + mg) * 3 + tm) * 64) + lane) * 8 + i] = t3[tm];
            }
  std::vector<float> packed(out.size() / 2);
  std::memcpy(packed.data(), out.data(), out.size() * 2);
  return packed;
}

std::vector<float> pack_conv_weights(const float *W, int Co, int Ci, int ntaps, int Ci_pad, int CK, int NB) {
  const int TN = 32 * NB, ntn = (Co + TN - 1) / TN, nch = Ci_pad / CK, K8 = CK / 8, nsteps = ntaps * K8;
  std::vector<float> out((size_t)ntn * nch * nsteps * NB * 64 * 4, 0.f);
  size_t o = 0;
  for (int nt = 0; nt < ntn; ++nt)
    for (int ch = 0; ch < nch; ++ch)
      for (int s = 0; s < nsteps; ++s) {
        const int t = s / K8, j = s % K8;
        for (int nb = 0; nb < NB; ++nb)
          for (int lane = 0; lane < 64; ++lane)
            for (int jj = 0; jj < 4; ++jj, ++o) {
              const int co = nt * TN + nb * 32 + (lane & 31);
              const int ci = ch * CK + 8 * j + 4 * (lane >> 5) + jj;
              if (co < Co && ci < Ci) out[o] = W[((size_t)co * Ci + ci) * ntaps + t];
            }
      }
  return out;
}

// Winograd F(2x2, 3x3) weights over the in-plane taps (dy, dx), one 4x4 transform G g G^T per (co, ci, dz):
//   G = (1,0,0), (1/2,1/2,1/2), (1/2,-1/2,1/2), (0,0,1).
// Layout (cm_conv_wino.hip): [n tile][16-channel chunk][xi_y][step = (dz*2 + k8)*4 + xi_x][lane][jj] with
// lane = 32*hh + (co % 32), ci = chunk*16 + 8*k8 + 4*hh + jj.  `wi` is the internal tap order [Co][Ci][27].
// With `ii` (global indices of the taps) the same walk emits, per packed element, the 9 (index, coefficient)
// terms for the device-side re-pack after an optimizer step.
void pack_wino(const std::vector<float> *wi, const std::vector<int> *ii, int Co, int Ci, int Ci_pad, std::vector<float> *out,
               std::vector<int> *oidx, std::vector<float> *ocoef) {
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  const int ntn = (Co + 31) / 32, nch = Ci_pad / 16;
  const size_t total = (size_t)ntn * nch * 4 * 24 * 64 * 4;
  if (out) out->assign(total, 0.f);
  if (oidx) { oidx->assign(total * 9, -1); ocoef->assign(total * 9, 0.f); }
  for (int co = 0; co < Co; ++co)
    for (int ci = 0; ci < Ci; ++ci)
      for (int dz = 0; dz < 3; ++dz)
        for (int xy = 0; xy < 4; ++xy)
          for (int xx = 0; xx < 4; ++xx) {
            const int nt = co / 32, r = co % 32, chunk = ci / 16, k8 = (ci % 16) / 8, hh = (ci % 8) / 4, jj = ci % 4;
            const size_t o = ((((((size_t)nt * nch + chunk) * 4 + xy) * 24 + (dz * 2 + k8) * 4 + xx) * 64) + hh * 32 + r) * 4 + jj;
            double acc = 0;
            for (int dy = 0; dy < 3; ++dy)
              for (int dx = 0; dx < 3; ++dx) {
                const size_t t = ((size_t)co * Ci + ci) * 27 + (dz * 3 + dy) * 3 + dx;
                const double c = G[xy][dy] * G[xx][dx];
                if (wi) acc += c * (double)(*wi)[t];
                if (oidx && c != 0.0) { (*oidx)[o * 9 + dy * 3 + dx] = (*ii)[t]; (*ocoef)[o * 9 + dy * 3 + dx] = (float)c; }
              }
            if (out) (*out)[o] = (float)acc;
          }
}

// IEEE binary16 bits of a float (round to nearest even; overflow -> infinity, like a device cast)
uint16_t f32_to_f16_bits(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const int32_t exp = (int32_t)((x >> 23) & 0xff) - 127 + 15;
  uint32_t man = x & 0x7fffffu;
  if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
  if (exp >= 31) return (uint16_t)(sign | 0x7c00u);
  if (exp <= 0) {
    if (exp < -10) return (uint16_t)sign;
    man |= 0x800000u;
    const int shift = 14 - exp;
    uint32_t h = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1))) ++h;
    return (uint16_t)(sign | h);
  }
  uint32_t h = ((uint32_t)exp << 10) | (man >> 13);
  const uint32_t rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
  return (uint16_t)(sign | h);
}

// f16 packing of the Winograd weights (cm_conv_wino.hip, F16): [n tile][chunk][xi_y][dz][xi_x][lane][8 halves],
// lane = 32*hh + co % 32, ci = chunk*16 + 8*hh + j.  Returned as floats holding two halves each (upload helper).
std::vector<float> pack_wino_f16(const std::vector<float> &wi, int Co, int Ci, int Ci_pad) {
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  const int ntn = (Co + 31) / 32, nch = Ci_pad / 16;
  std::vector<uint16_t> out((size_t)ntn * nch * 4 * 3 * 4 * 64 * 8, 0);
  for (int co = 0; co < Co; ++co)
    for (int ci = 0; ci < Ci; ++ci)
      for (int dz = 0; dz < 3; ++dz)
        for (int xy = 0; xy < 4; ++xy)
          for (int xx = 0; xx < 4; ++xx) {
            const int nt = co / 32, r = co % 32, chunk = ci / 16, hh = (ci % 16) / 8, j = ci % 8;
            double acc = 0;
            for (int dy = 0; dy < 3; ++dy)
              for (int dx = 0; dx < 3; ++dx) acc += G[xy][dy] * G[xx][dx] * (double)wi[((size_t)co * Ci + ci) * 27 + (dz * 3 + dy) * 3 + dx];
            const size_t o = (((((((size_t)nt * nch + chunk) * 4 + xy) * 3 + dz) * 4 + xx) * 64) + hh * 32 + r) * 8 + j;
            out[o] = f32_to_f16_bits((float)acc);
          }
  std::vector<float> packed(out.size() / 2);
  std::memcpy(packed.data(), out.data(), out.size() * 2);
  return packed;
}

// Six-term bf16 form of the Winograd layers (conv_wino_p_kernel<..., B6>): the fp32 fragments of pack_wino
// ([n tile][chunk][xi_y][step = (dz * 2 + k8) * 4 + xi_x][lane][4], ci = chunk * 16 + 8 k8 + 4 hh + jj) split exactly into three
// bf16 terms and regrouped as [n tile][chunk][xi_y][dz][xi_x][term][lane][8 bf16], ci = chunk * 16 + 8 hh + j.  The device
// re-derives the same thing after an optimizer step (wino_b6_repack_kernel): one definition, two places -- the self-test
// compares them element by element.
// h2_scale > 0: the h2 form instead -- f16 hi / mid of w * h2_scale in the first two term slots (same layout, third slot zero)
std::vector<float> pack_wino_b6(const std::vector<float> &ww, float h2_scale = 0.f) {
  std::vector<uint16_t> out(ww.size() * 3, 0);
  for (size_t i = 0; i < ww.size(); ++i) {
    const int jj = (int)(i & 3), lane = (int)((i >> 2) & 63);
    size_t q = i >> 8;
    const int step = (int)(q % 24); q /= 24;
    const int xy = (int)(q & 3);
    const size_t tc = q >> 2;
    const int xx = step & 3, k8 = (step >> 2) & 1, dz = step >> 3;
    const int r = lane & 31, hs = lane >> 5, cl = 8 * k8 + 4 * hs + jj, hd = cl >> 3, j = cl & 7;
    uint16_t t3[3];
    if (h2_scale > 0.f) f16_split2(ww[i], h2_scale, t3);
    else bf16_split3(ww[i], t3);
    for (int tm = 0; tm < 3; ++tm)
      out[(((((((tc * 4 + xy) * 3 + dz) * 4 + xx) * 3 + tm) * 64) + 32 * hd + r) * 8) + j] = t3[tm];
  }
  std::vector<float> packed(out.size() / 2);
  std::memcpy(packed.data(), out.data(), out.size() * 2);
  return packed;
}

// f16 fragments of a 1x1x1 conv for conv1x1_f16_kernel: [n tile][16-channel group][block][lane 64][8 halves],
// lane (r, h) of block nb holds W[co = (nt NB + nb) 32 + r][ci = 16 g + 8 h + j]; W is [Co][Ci]
std::vector<float> pack_1x1_f16(const float *W, int Co, int Ci, int NB) {
  const int TN = 32 * NB, ntn = (Co + TN - 1) / TN, ng = Ci / 16;
  std::vector<uint16_t> out((size_t)ntn * ng * NB * 64 * 8, 0);
  size_t o = 0;
  for (int nt = 0; nt < ntn; ++nt)
    for (int g = 0; g < ng; ++g)
      for (int nb = 0; nb < NB; ++nb)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j, ++o) {
            const int co = (nt * NB + nb) * 32 + (lane & 31), ci = 16 * g + 8 * (lane >> 5) + j;
            if (co < Co) out[o] = f32_to_f16_bits(W[(size_t)co * Ci + ci]);
          }
  std::vector<float> packed(out.size() / 2);
  std::memcpy(packed.data(), out.data(), out.size() * 2);
  return packed;
}

int pick_ck(int C0, int C1) {
  for (int ck : {32, 16, 8})
    if (C0 % ck == 0 && C1 % ck == 0) return ck;
  return 0;
}

// Measured tile choices for the layer shapes of the reference configurations (tools/tune_tiles.py on
// one MI355X at B = 64, standalone launches): the analytic score below orders candidates well within a
// shape class but not across accumulator blockings.  Keyed by the layer shape only -- never by the batch --
// so the geometry, and with it every rounding, stays independent of how a batch is sharded.
struct TunedTile { int ntaps, stride, par, Ci, Co, Zo, Yo, Xo, NB, MB, bz, by, bx; };
const TunedTile kTunedTiles[] = {
#include "cm_tuned_tiles.inc"
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};

const TunedTile *find_tuned(const cm::ConvArgs &a, int NB /* 0: any */) {
  static const bool off = cm::diag_env("CM_NO_TUNED") != nullptr;
  if (off) return nullptr;
  for (const TunedTile *t = kTunedTiles; t->ntaps; ++t)
    if (t->ntaps == a.ntaps && t->stride == a.stride && t->par == a.par && t->Ci == a.C0 + a.C1 && t->Co == a.Co &&
        t->Zo == a.Zo && t->Yo == a.Yo && t->Xo == a.Xo && (NB == 0 || t->NB == NB))
      return t;
  return nullptr;
}

// Tile geometry for one conv at batch B: choose the output box (bs,bz,by,bx) and
// the per-wave accumulator blocking MB (NB is fixed by the packed weights).
// Tile geometry is chosen for a fixed reference batch, never for the batch at hand: the
// fused GroupNorm statistics are summed per tile, so a geometry that changed with the
// batch size would make sample i of a shard differ in the last bits from sample i of the
// unsharded batch (SURVEY.md section 8e asks for bit-identical shards).
constexpr int TUNE_BATCH = 64;

// First conv (cm_conv_io.hip): tiles are bz x by planes of FULL x-rows; op.MB = 32-row blocks per tile
// (= statistics slots per tile).  Score: fill of the last block, balance over the 4 waves, halo overhead,
// enough tiles for the chip at the reference batch.
void pick_tile_first(Op &op) {
  cm::ConvArgs &a = op.ca;
  double best = -1;
  int bbz = 1, bby = 1;
  a.bs = 1; a.bx = a.Xo;
  for (int bz = 1; bz <= a.Zo; ++bz)
    for (int by = 1; by <= a.Yo; ++by) {
      if (a.Zo % bz || a.Yo % by) continue;
      a.bz = bz; a.by = by;
      const int nbox = bz * by * a.Xo, nblk = (nbox + 31) / 32;
      if (nblk > 16 || cm::conv_first_lds(a, op.first_cin) > 48 * 1024) continue;
      const double tiles = (double)(a.Zo / bz) * (a.Yo / by) * TUNE_BATCH;
      if ((a.Zo / bz) * (a.Yo / by) * nblk > MAX_SLOTS) continue;
      const double eff = (double)nbox / (32.0 * nblk);
      const double bal = (double)nblk / (4.0 * ((nblk + 3) / 4));
      const double halo = (double)nbox / ((bz + 2.0) * (by + 2.0) * (a.Xo + 2.0));
      const double score = eff * bal * (0.5 + 0.5 * halo) * std::min(1.0, tiles / 512.0);
      if (score > best) { best = score; bbz = bz; bby = by; }
    }
  a.bz = bbz; a.by = bby;
  a.ntz = a.Zo / bbz; a.nty = a.Yo / bby; a.ntx = 1;
  op.MB = cm::conv_first_blocks(a);
}

void pick_tile(Op &op, int B) {
  cm::ConvArgs &a = op.ca;
  const int NB = op.NB;
  static const int force_mb = cm::diag_env("CM_FORCE_MB") ? atoi(cm::diag_env("CM_FORCE_MB")) : 0;
  const int osd = a.par ? 2 : 1;  // parity mode tiles the low-resolution source grid
  const int Zo = a.Zo / osd, Yo = a.Yo / osd, Xo = a.Xo / osd;
  const int vox = Zo * Yo * Xo;
  // largest accumulator blocking that still fits 256 VGPRs (2 waves/SIMD) without spilling;
  // the register-ring fast path (CK == 32) carries more live state than the generic one
  const bool fastp = a.CK == 32 && (a.ntaps == 27 || a.ntaps == 8);
  const int max_blk = fastp ? ((NB == 1) ? 5 : (NB == 2 ? 2 : 1)) : ((NB == 1) ? 8 : (NB == 2 ? 4 : 2));
  if (const TunedTile *t = find_tuned(a, NB)) {
    const int osdt = a.par ? 2 : 1;
    a.bs = 1; a.bz = t->bz; a.by = t->by; a.bx = t->bx;
    if (cm::conv_variant_exists(t->MB, NB) && (t->bz * t->by * t->bx + 31) / 32 == t->MB && cm::conv_lds_bytes(a, t->MB, NB) <= 80 * 1024 &&
        (!op.small_n || !(t->MB & (t->MB - 1)))) {
      op.MB = t->MB;
      a.ntz = (a.Zo / osdt + a.bz - 1) / a.bz; a.nty = (a.Yo / osdt + a.by - 1) / a.by; a.ntx = (a.Xo / osdt + a.bx - 1) / a.bx;
      op.tuned_B = B;
      return;
    }
  }
  double best = -1;
  int bbs = 1, bbz = 1, bby = 1, bbx = 1, bMB = 1;
  const int ntn = (a.Co + 32 * NB - 1) / (32 * NB);
  const int max_bs = (vox <= 64 && !op.stat_act && !op.small_n) ? 4 : 1;  // fused statistics need one sample per tile
  for (int bs = 1; bs <= max_bs; ++bs)
    for (int bz = 1; bz <= Zo; ++bz)
      for (int by = 1; by <= Yo; ++by)
        for (int bx = 1; bx <= Xo; ++bx) {
          const int nbox = bs * bz * by * bx;
          const int MB = (nbox + 31) / 32;
          if (MB > max_blk) continue;
          if (op.small_n && (MB & (MB - 1))) continue;  // its thread groups need 256 % (32 MB) == 0
          if (force_mb && MB != force_mb && vox >= 128) continue;
          a.bs = bs; a.bz = bz; a.by = by; a.bx = bx;
          const size_t lds = cm::conv_lds_bytes(a, MB, NB);
          if (lds > 64 * 1024) continue;
          const long ntz = (Zo + bz - 1) / bz, nty = (Yo + by - 1) / by, ntx = (Xo + bx - 1) / bx, nts = (B + bs - 1) / bs;
          const double tiles = (double)ntz * nty * ntx * nts * ntn * (a.par ? 8 : 1);
          const double util = (double)vox * B * ntn / (tiles * 32.0 * MB);
          // work per tile in accumulator blocks; balance over 256 CUs
          const double per_cu = std::ceil(tiles / 256.0);
          const double balance = tiles / (per_cu * 256.0);
          // halo overhead (staging) and register-occupancy preference
          const double hv = (double)bs * ((bz - 1) * a.stride + a.td) * ((by - 1) * a.stride + a.td) *
                            ((bx - 1) * a.stride + a.td);
          const double halo = 1.0 / (1.0 + 0.02 * hv / (32.0 * MB));
          const double occ = (MB * NB <= 4) ? 1.0 : 0.93;
          const double amort = 1.0 - 0.06 / (MB * NB);  // larger blocks amortise loads/epilogue
          const double score = util * std::min(1.0, 0.25 + balance) * halo * occ * amort;
          if (score > best) { best = score; bbs = bs; bbz = bz; bby = by; bbx = bx; bMB = MB; }
        }
  a.bs = bbs; a.bz = bbz; a.by = bby; a.bx = bbx;
  op.MB = bMB;
  a.ntz = (Zo + bbz - 1) / bbz; a.nty = (Yo + bby - 1) / bby; a.ntx = (Xo + bbx - 1) / bbx;
  op.tuned_B = B;
}

struct ConvSpec {
  const Act *s0;
  const Act *s1 = nullptr;
  const float *gn = nullptr;
  int silu = 0;
  std::string wname, bname;
  int ntaps = 27, stride = 1, ups = 0;
  const float *temb = nullptr;
  const Act *resid = nullptr;
  Act *out = nullptr;
  int Co = 0;
  int ci_valid = -1;  // valid input channels of the reference weight (first conv: 3 of 8)
  bool stats = false; // produce the GroupNorm statistics of `out` in the epilogue
  int pm_off = -1;    // Dropout3d mask slice of the input (conv_2 of a ResnetBlock)
  const Act *skip0 = nullptr, *skip1 = nullptr;  // block input whose match_input conv may be fused in
  std::string skip_w, skip_b;
};

// f16 fragments of the direct f16 kernel (cm_conv_f16.hip): [Co/(32 NB)][Ci/16][taps][NB][lane][8 halves] with
// co = 32 NB nt + 32 nb + lane % 32, ci = 16 c + 8 (lane / 32) + j; `w` is [Co][Ci][taps] (taps = 27 internal order, or 1)
std::vector<float> pack_f16d(const float *w, int Co, int Ci, int taps, int NB) {
  const int ntn = Co / (32 * NB), nc = Ci / 16;
  std::vector<uint16_t> out((size_t)ntn * nc * taps * NB * 64 * 8, 0);
  size_t o = 0;
  for (int nt = 0; nt < ntn; ++nt)
    for (int c = 0; c < nc; ++c)
      for (int t = 0; t < taps; ++t)
        for (int nb = 0; nb < NB; ++nb)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j, ++o) {
              const int co = nt * 32 * NB + nb * 32 + (lane & 31), ci = 16 * c + 8 * (lane >> 5) + j;
              out[o] = f32_to_f16_bits(w[((size_t)co * Ci + ci) * taps + t]);
            }
  std::vector<float> packed(out.size() / 2);
  std::memcpy(packed.data(), out.data(), out.size() * 2);
  return packed;
}

// bf16 x 3 fragments of the direct six-term kernel (cm_conv_b6d.hip): [Co/(32 NB)][Ci/16][taps][NB][term hi / mid / lo][lane][8 bf16]
// with co = 32 NB nt + 32 nb + lane % 32, ci = 16 c + 8 (lane / 32) + j; `w` is [Co][Ci][taps] (taps = 27 internal order, or 1).
// The device re-derives them from the reference-layout master weights after an optimizer step (b6d_repack_kernel).
std::vector<float> pack_b6d(const float *w, int Co, int Ci, int taps, int NB) {
  const int ntn = Co / (32 * NB), nc = Ci / 16;
  std::vector<uint16_t> out((size_t)ntn * nc * taps * NB * 3 * 64 * 8, 0);
  for (int nt = 0; nt < ntn; ++nt)
    for (int c = 0; c < nc; ++c)
      for (int t = 0; t < taps; ++t)
        for (int nb = 0; nb < NB; ++nb)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int co = nt * 32 * NB + nb * 32 + (lane & 31), ci = 16 * c + 8 * (lane >> 5) + j;
              uint16_t t3[3];
              bf16_split3(w[((size_t)co * Ci + ci) * taps + t], t3);
              for (int tm = 0; tm < 3; ++tm)
                out[(((((((size_t)nt * nc + c) * taps + t) * NB + nb) * 3 + tm) * 64) + lane) * 8 + j] = t3[tm];
            }
  std::vector<float> packed(out.size() / 2);
  std::memcpy(packed.data(), out.data(), out.size() * 2);
  return packed;
}

// Weights of the whole-sample quarter-resolution kernel (cm_conv_qr.hip): [Co/32][g = k8*9 + dy*3 + dx][dz][lane][jj]
// with co = 32 nt + lane % 32, ci = 8 k8 + 4 (lane / 32) + jj; `wi` in the internal tap order [Co][Ci][(dz*3 + dy)*3 + dx].
std::vector<float> pack_qr(const std::vector<float> &wi, int Co, int Ci) {
  const int ntn = Co / 32, K8 = Ci / 8, ng = 9 * K8;
  std::vector<float> out((size_t)ntn * ng * 3 * 64 * 4, 0.f);
  size_t o = 0;
  for (int nt = 0; nt < ntn; ++nt)
    for (int g = 0; g < ng; ++g)
      for (int dz = 0; dz < 3; ++dz)
        for (int lane = 0; lane < 64; ++lane)
          for (int jj = 0; jj < 4; ++jj, ++o) {
            const int k8 = g / 9, t9 = g % 9, dy = t9 / 3, dx = t9 % 3;
            const int co = nt * 32 + (lane & 31), ci = 8 * k8 + 4 * (lane >> 5) + jj;
            out[o] = wi[((size_t)co * Ci + ci) * 27 + (dz * 3 + dy) * 3 + dx];
          }
  return out;
}
// Six-term bf16 form of conv_qr2 (B6): the fp32 fragments of pack_qr split exactly into three bf16 terms and regrouped by wave
// (wave w owns the channels [w Ci/8, (w+1) Ci/8), padded with zeros to whole 16-channel steps):
// [Co/32][wave 8][step][tap 9][dz][term][lane][8 bf16], ci = wave * Ci/8 + 16 step + 8 hh + j.  `wq` is pack_qr's output.
// The device re-derives it after an optimizer step with the same index arithmetic (qr_b6_repack_kernel).
std::vector<float> pack_qr_b6(const std::vector<float> &wq, int Co, int Ci, float h2_scale = 0.f) {
  const int ntn = Co / 32, K8 = Ci / 8, ng = 9 * K8, cw = Ci / 8, nsw = (cw + 15) / 16;
  std::vector<uint16_t> out((size_t)ntn * 8 * nsw * 9 * 3 * 3 * 64 * 8, 0);
  for (size_t i = 0; i < wq.size(); ++i) {
    const int jj = (int)(i & 3), lane = (int)((i >> 2) & 63);
    size_t q = i >> 8;
    const int dz = (int)(q % 3); q /= 3;
    const int g = (int)(q % ng);
    const int nt = (int)(q / ng);
    const int k8 = g / 9, t9 = g % 9, ci = 8 * k8 + 4 * (lane >> 5) + jj, r = lane & 31;
    const int wv = ci / cw, cl = ci % cw, st = cl / 16, hd = (cl % 16) / 8, j = cl % 8;
    uint16_t t3[3];
    if (h2_scale > 0.f) f16_split2(wq[i], h2_scale, t3);
    else bf16_split3(wq[i], t3);
    for (int tm = 0; tm < 3; ++tm)
      out[(((((((size_t)(nt * 8 + wv) * nsw + st) * 9 + t9) * 3 + dz) * 3 + tm) * 64) + 32 * hd + r) * 8 + j] = t3[tm];
  }
  std::vector<float> packed(out.size() / 2);
  std::memcpy(packed.data(), out.data(), out.size() * 2);
  return packed;
}
// its fused 1x1x1 skip weights: [Co/32][Cs/8][lane][jj]; `w2` is [Co][Cs]
std::vector<float> pack_qr_skip(const float *w2, int Co, int Cs) {
  const int ntn = Co / 32, ngs = Cs / 8;
  std::vector<float> out((size_t)ntn * ngs * 64 * 4, 0.f);
  size_t o = 0;
  for (int nt = 0; nt < ntn; ++nt)
    for (int gs = 0; gs < ngs; ++gs)
      for (int lane = 0; lane < 64; ++lane)
        for (int jj = 0; jj < 4; ++jj, ++o)
          out[o] = w2[(size_t)(nt * 32 + (lane & 31)) * Cs + 8 * gs + 4 * (lane >> 5) + jj];
  return out;
}

// ---- "h2" arithmetic (cm_kernels.h: cm_split2_f16): operand range management ---------------------------------------------------
// f16 has 5 exponent bits.  Weights: packed as w * 2^k with max |w| 2^k in [4096, 8192) (their mid terms ~ 2^-11 of that stay normal
// numbers; 8x headroom below 65504); the kernel multiplies its fp32 accumulators by 2^-k (exact).  Activations: a GroupNorm output
// satisfies |z| < sqrt(n) for a group of n elements, so |SiLU(gamma z + beta)| <= sqrt(n) max|gamma| + max|beta|, times `gain` for a
// linear input transform (4 for the Winograd B^T d B: sums of four values).  A layer whose bound exceeds 32000 keeps the six-term
// bf16 form (bf16 has the exponent range of fp32).
static float h2_wscale(const float *w, size_t n) {
  float mx = 0.f;
  for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::fabs(w[i]));
  if (!(mx > 0.f) || !std::isfinite(mx)) return 0.f;
  int e = 0;
  (void)std::frexp(mx, &e);                        // mx = f * 2^e, f in [0.5, 1)
  return std::ldexp(1.0f, 13 - e);                 // mx * 2^(13 - e) in [4096, 8192)
}
static bool h2_bound_ok(const std::vector<float> &g, const std::vector<float> &b, const Op &gop, double gain);
static bool h2_act_bounded(const cm_model *m, const Op &gop, double gain) {
  return h2_bound_ok(P(m, gop.gname).host, P(m, gop.bename).host, gop, gain);
}
static bool h2_bound_ok(const std::vector<float> &g, const std::vector<float> &b, const Op &gop, double gain) {
  double gm = 0, bm = 0;
  for (float v : g) gm = std::max(gm, (double)std::fabs(v));
  for (float v : b) bm = std::max(bm, (double)std::fabs(v));
  const int Ct = gop.g0->C + (gop.g1 ? gop.g1->C : 0);
  const double n = (double)(Ct / GN_GROUPS) * gop.g0->V();
  const double bound = (std::sqrt(n) * gm + bm) * gain;
  return std::isfinite(bound) && bound <= 32000.0;
}

// h2 fragments of one layer from its weights in the REFERENCE layout (load time: add_conv; after training: refresh_h2).  Return the
// weight scale 2^k (0: none -- all-zero or non-finite weights).
static float h2_pack_wino(const float *w_ref, int Co, int Ci_ref, int Ci_pad, std::vector<float> *out) {
  const std::vector<float> wi = to_internal_taps(w_ref, Co, Ci_ref, 27);
  std::vector<float> ww;
  pack_wino(&wi, nullptr, Co, Ci_ref, Ci_pad, &ww, nullptr, nullptr);
  const float ws = h2_wscale(ww.data(), ww.size());
  if (ws > 0.f) *out = pack_wino_b6(ww, ws);
  return ws;
}
static float h2_pack_qr(const float *w_ref, int Co, int Ci_ref, std::vector<float> *out) {
  const std::vector<float> wq = pack_qr(to_internal_taps(w_ref, Co, Ci_ref, 27), Co, Ci_ref);
  const float ws = h2_wscale(wq.data(), wq.size());
  if (ws > 0.f) *out = pack_qr_b6(wq, Co, Ci_ref, ws);
  return ws;
}
static float h2_pack_ups(const float *w_ref, int Co, int Ci_ref, std::vector<float> *out) {
  const std::vector<float> wp = parity_weights(to_internal_taps(w_ref, Co, Ci_ref, 27), Co, Ci_ref);
  const size_t per = (size_t)Co * Ci_ref * 8;
  const float ws = h2_wscale(wp.data(), wp.size());
  if (ws > 0.f) {
    out->clear();
    for (int p8 = 0; p8 < 8; ++p8) {
      const std::vector<float> one = pack_ups_b6(wp.data() + p8 * per, Co, Ci_ref, ws);
      out->insert(out->end(), one.begin(), one.end());
    }
  }
  return ws;
}

// Which Winograd-eligible layers take the direct six-term kernel instead.  Round 4, same-box A/B on the full-resolution layers
// (B = 64, ATC): 82 / 84 / 208 / 109 / 142 / 99 us against the six-term Winograd kernel's 55 / 62 / 144 / 76 / 98 / 67 (step 1.563
// vs 1.381 ms) -- its phases add up (skeleton 23 + matrix 50 + epilogue 10 us on the 32 -> 32 layer), see DESIGN section 6 --
// so NONE by default; CM_B6D=full (Co = 32, >= 8 planes) / all under CM_DIAG=1 select it for A/B runs and the parity tests.
bool b6d_wanted(int Co, int Z, int Y, int X) {
  (void)Y; (void)X;
  if (const char *e = cm::diag_env("CM_B6D")) {
    if (!strcmp(e, "all")) return true;
    if (!strcmp(e, "full")) return Co == 32 && Z >= 8;
  }
  return false;
}

int add_conv(cm_model *m, const ConvSpec &s) {
  Op op;
  op.kind = OP_CONV;
  op.cls = s.ntaps == 27 ? K_CONV3 : K_CONV1;
  cm::ConvArgs &a = op.ca;
  a.src0 = s.s0->d; a.C0 = s.s0->C;
  a.src1 = s.s1 ? s.s1->d : nullptr; a.C1 = s.s1 ? s.s1->C : 0;
  a.gn = s.gn; a.silu = s.silu;
  a.Zs = s.s0->Z; a.Ys = s.s0->Y; a.Xs = s.s0->X;
  a.Zo = s.out->Z; a.Yo = s.out->Y; a.Xo = s.out->X;
  a.ntaps = s.ntaps; a.stride = s.stride; a.ups = s.ups;
  a.td = s.ntaps == 27 ? 3 : 1; a.par = 0; a.wpar_stride = 0;
  const bool parity = s.ups && s.ntaps == 27 && !cm::diag_env("CM_NO_PARITY_UPCONV");
  if (parity) { a.ups = 0; a.par = 1; a.td = 2; a.ntaps = 8; }
  a.out = s.out->d; a.out_cs = s.out->C; a.Co = s.Co;
  a.temb = s.temb; a.temb_stride = m->nproj; a.tidx = m->tbuf;
  a.resid = s.resid ? s.resid->d : nullptr; a.res_cs = s.resid ? s.resid->C : 0;
  a.CK = pick_ck(a.C0, a.C1);
  if (s.ntaps == 1) {
    // 1x1x1 convs have no halo: stage as many channels per pass as the sources allow
    // (a chunk may not straddle the concat boundary), so a whole K = Ci contraction needs
    // one or two staging rounds instead of Ci/32.
    for (int ck : {256, 128, 64})
      if (a.C0 % ck == 0 && a.C1 % ck == 0) { a.CK = ck; break; }
  }
  if (!a.CK) return fail("conv %s: channel counts %d/%d not multiples of 8", s.wname.c_str(), a.C0, a.C1);
  a.nch0 = a.C0 / a.CK; a.nch1 = a.C1 / a.CK;
  op.NB = s.Co > 32 ? 2 : 1;
  // full- and half-resolution stride-1 3x3x3 layers: Winograd F(2x2,3x3) over (Y, X) -- 2.25x fewer matrix
  // instructions (cm_conv_wino.hip); its workgroups own one 32-channel output tile each
  {
    int wz = 0, wy = 0, wx = 0;
    op.wino = s.ntaps == 27 && s.stride == 1 && !s.ups && a.C0 % 16 == 0 && a.C1 % 16 == 0 && s.Co % 32 == 0 && s.Co == s.out->C &&
              s.ci_valid < 0 && s.out->V() > 64 && cm::conv_wino_pick(s.out->Z, s.out->Y, s.out->X, &wz, &wy, &wx) && !cm::diag_env("CM_NO_WINO");
  }
  // tiny-spatial layers are overhead-bound, not throughput-bound: fewer, fatter workgroups
  // (all 128 output channels per workgroup, K split over workgroups) amortise the per-workgroup
  // fixed costs over 4x the matrix work
  // (NB = 4 "fat" tiles -- all 128 output channels per workgroup -- spill ~80 VGPRs on the register-ring path and
  // measured no better than two NB = 2 workgroups: opt-in only)
  if (s.ntaps == 27 && s.out->V() <= 64 && s.Co % 128 == 0 && cm::diag_env("CM_FAT_TILES")) op.NB = 4;
  if (s.ntaps == 27 && s.out->V() <= 64 && cm::diag_env("CM_QR_NB")) op.NB = atoi(cm::diag_env("CM_QR_NB"));
  if (op.wino) op.NB = 1;
  if (s.ntaps == 27 && s.out->V() > 64 && !op.wino) {
    if (const TunedTile *t = find_tuned(a, 0)) op.NB = t->NB;
    // tuner policies (tools/tune_tiles.py): N blocking of the 64- / 128-channel 3x3x3 layers
    if (s.Co == 64 && cm::diag_env("CM_NB64")) op.NB = atoi(cm::diag_env("CM_NB64"));
    if (s.Co == 128 && cm::diag_env("CM_NB128")) op.NB = atoi(cm::diag_env("CM_NB128"));
  }
  const Param &w = P(m, s.wname);
  const Param &b = P(m, s.bname);
  const int Ci_ref = (int)w.shape[1];
  const int Ci_pad = a.C0 + a.C1;
  if (Ci_ref > Ci_pad || (s.ci_valid < 0 && Ci_ref != Ci_pad))
    return fail("conv %s: weight has %d input channels, sources provide %d", s.wname.c_str(), Ci_ref, Ci_pad);
  const std::vector<float> wi = to_internal_taps(w.host.data(), (int)w.shape[0], Ci_ref, s.ntaps);
  std::vector<float> wf;
  if (parity) {
    const std::vector<float> wp = parity_weights(wi, (int)w.shape[0], Ci_ref);
    const size_t per = (size_t)w.shape[0] * Ci_ref * 8;
    for (int p8 = 0; p8 < 8; ++p8) {
      std::vector<float> one = pack_conv_weights(wp.data() + p8 * per, (int)w.shape[0], Ci_ref, 8, Ci_pad, a.CK, op.NB);
      a.wpar_stride = (long long)one.size();
      wf.insert(wf.end(), one.begin(), one.end());
    }
    if (m->precision == CM_PRECISION_F16 && a.CK == 32) {   // reduced-precision plan: the same fragments as f16
      std::vector<float> wf16;
      for (int p8 = 0; p8 < 8; ++p8) {
        std::vector<float> one = pack_conv_weights_f16(wp.data() + p8 * per, (int)w.shape[0], Ci_ref, 8, Ci_pad, a.CK, op.NB);
        op.wpar_stride16 = (long long)one.size();
        wf16.insert(wf16.end(), one.begin(), one.end());
      }
      if (upload(m, wf16, &op.d_wfrag16)) return 1;
    }
  } else {
    wf = pack_conv_weights(wi.data(), (int)w.shape[0], Ci_ref, s.ntaps, Ci_pad, a.CK, op.NB);
    // reduced-precision plan: 1x1x1 convs without output statistics (attention in-projection, unfused skip convs) on f16 operands
    if (m->precision == CM_PRECISION_F16 && s.ntaps == 1 && s.stride == 1 && !s.ups && !s.stats && !s.temb && Ci_ref == Ci_pad &&
        a.C0 % 16 == 0 && a.C1 % 16 == 0 && !cm::diag_env("CM_NO_1X1_F16") &&
        upload(m, pack_1x1_f16(wi.data(), (int)w.shape[0], Ci_ref, op.NB), &op.d_w1x1_16))
      return 1;
  }
  // upsample convs: the source tile staged once for four parity classes (cm_conv_ups.hip) when a tile fits
  if (parity && a.CK == 32 && !s.s1 && !s.gn && !s.temb && !s.resid && s.Co % 32 == 0 && s.Co == s.out->C && !cm::diag_env("CM_NO_UPS") &&
      cm::conv_ups_pick(a.Zs, a.Ys, a.Xs, &op.ups_tz, &op.ups_ty, &op.ups_tx, &op.ups_mbw, &op.ups_planes) &&
      (a.Zs / op.ups_tz) * (a.Ys / op.ups_ty) * (a.Xs / op.ups_tx) * 8 * op.ups_mbw <= MAX_SLOTS)   // (its statistics slots must fit)
    op.ups = true;
  if (op.ups && m->precision != CM_PRECISION_F16 && Ci_ref == Ci_pad && Ci_ref % 32 == 0 && !cm::diag_env("CM_NO_UPS_B6")) {
    const std::vector<float> wp = parity_weights(wi, (int)w.shape[0], Ci_ref);
    const size_t per = (size_t)w.shape[0] * Ci_ref * 8;
    std::vector<float> wb6;
    for (int p8 = 0; p8 < 8; ++p8) {
      const std::vector<float> one = pack_ups_b6(wp.data() + p8 * per, (int)w.shape[0], Ci_ref);
      op.wups_b6_stride = (long long)one.size();
      wb6.insert(wb6.end(), one.begin(), one.end());
    }
    if (upload(m, wb6, &op.d_wups_b6)) return 1;
    // default plan: h2 with a per-sample scale taken from the source tensor's slot statistics (cm_conv_ups.hip, PREC = 4) -- the
    // source is a block output with statistics (every conv_2 / attention output carries them)
    if (m->precision == CM_PRECISION_F32 && !cm::diag_env("CM_NO_H2") && !cm::diag_env("CM_NO_UPS_H2") && s.s0->part) {
      std::vector<float> wh2;
      const float ws = h2_pack_ups(w.host.data(), (int)w.shape[0], Ci_ref, &wh2);
      if (ws > 0.f) {
        if (upload(m, wh2, &op.d_wups_h2)) return 1;
        op.h2_oscale = 1.f / ws;
      }
    }
  }
  if (op.ups && m->precision == CM_PRECISION_F16 && Ci_ref == Ci_pad && Ci_ref % 32 == 0 && !cm::diag_env("CM_NO_UPS_F16")) {
    const std::vector<float> wp = parity_weights(wi, (int)w.shape[0], Ci_ref);
    const size_t per = (size_t)w.shape[0] * Ci_ref * 8;
    std::vector<float> w16;
    for (int p8 = 0; p8 < 8; ++p8) {
      const std::vector<float> one = pack_ups_f16(wp.data() + p8 * per, (int)w.shape[0], Ci_ref);
      op.wups16_stride = (long long)one.size();
      w16.insert(w16.end(), one.begin(), one.end());
    }
    if (upload(m, w16, &op.d_wups16)) return 1;
  }
  // the UNet's last conv (base -> C channels): vector-ALU kernel instead of a 32-wide MFMA tile
  if (s.ntaps == 27 && s.stride == 1 && !s.ups && s.Co <= 8 && !s.stats && !s.temb && !s.resid &&
      !cm::diag_env("CM_NO_SMALLN")) {
    op.small_n = true;
    op.small_nco = s.Co <= 4 ? 4 : 8;
    const int nco = op.small_nco, nch = Ci_pad / a.CK;
    std::vector<float> ws((size_t)nch * 27 * a.CK * nco, 0.f);
    for (int ch = 0; ch < nch; ++ch)
      for (int t = 0; t < 27; ++t)
        for (int ci = 0; ci < a.CK; ++ci)
          for (int co = 0; co < s.Co; ++co) {
            const int cig = ch * a.CK + ci;
            if (cig < Ci_ref) ws[(((size_t)ch * 27 + t) * a.CK + ci) * nco + co] = wi[((size_t)co * Ci_ref + cig) * 27 + t];
          }
    if (upload(m, ws, &op.d_wsmall)) return 1;
    // the same layer on the matrix core (round 4): 27 taps x 4 channels = 108 columns of a GEMM over the 32 input channels
    if (s.Co <= 4 && Ci_ref == 32 && Ci_pad == 32 && !s.s1 && !cm::diag_env("CM_NO_FIN") && cm::conv_fin_pick(s.out->Y, s.out->X, &op.fin_by, &op.fin_bx)) {
      if (upload(m, w.host, &op.d_wfin_src)) return 1;
      if (dev_alloc(m, (void **)&op.d_wfin, cm::CM_FIN_W_FLOATS * sizeof(float))) return 1;
      CM_HIP(cm::launch_fin_pack(op.d_wfin_src, op.d_wfin, s.Co, 0, m->stream));
      if (m->precision == CM_PRECISION_F16) {
        if (dev_alloc(m, (void **)&op.d_wfin16, cm::CM_FIN_W_FLOATS * sizeof(float))) return 1;
        CM_HIP(cm::launch_fin_pack(op.d_wfin_src, op.d_wfin16, s.Co, 1, m->stream));
      }
      CM_HIP(hipStreamSynchronize(m->stream));
      op.fin = true;
    }
  }
  if (op.wino) {
    std::vector<float> ww;
    pack_wino(&wi, nullptr, s.Co, Ci_ref, Ci_pad, &ww, nullptr, nullptr);
    if (upload(m, ww, &op.d_wwino)) return 1;
    if (m->precision == CM_PRECISION_F16 && upload(m, pack_wino_f16(wi, s.Co, Ci_ref, Ci_pad), &op.d_wwino16)) return 1;
    op.wwino_floats = (long long)ww.size();
    {
      int wz = 0, wy = 0, wx = 0;
      if (m->precision != CM_PRECISION_F16 && !cm::diag_env("CM_NO_WINO_B6") && cm::conv_wino_pick(s.out->Z, s.out->Y, s.out->X, &wz, &wy, &wx) &&
          cm::conv_wino_b6_ok(wz, wy, wx, s.Co, s.out->Z) && upload(m, pack_wino_b6(ww), &op.d_wwino_b6))
        return 1;
    }
    // reduced-precision plan: the direct f16 kernel replaces the Winograd one where its tiles fit (no transforms to pay for
    // when the matrix instruction is 16x faster)
    // (measured on the 24x72 grid, B = 32: 61 vs 70 us on the 32 -> 32 full-resolution layer, 136 vs 154 us on 96 -> 32, 72 vs 76 us
    // at half resolution; on two-plane grids the two-tile Winograd form stays: 24 vs 47 us)
    if (m->precision == CM_PRECISION_F16 && Ci_ref == Ci_pad && a.C0 % 16 == 0 && a.C1 % 16 == 0 && s.out->Z >= 4 && !cm::diag_env("CM_NO_F16D") &&
        cm::conv_f16d_pick(s.out->Z, s.out->Y, s.out->X, &op.f16d_bz, &op.f16d_by, &op.f16d_bx, &op.f16d_mbw)) {
      if (upload(m, pack_f16d(wi.data(), s.Co, Ci_ref, 27, s.Co % 64 == 0 ? 2 : 1), &op.d_w16d)) return 1;
      op.f16d = true;
    }
    // fp32 plan, full-resolution layers (Co = 32: one output tile, so the Winograd form has no second tile to share its U image with):
    // the direct six-term kernel -- the split paid once per staged element instead of once per frequency component, 1.5 instead of
    // 8 vector instructions per matrix instruction (cm_conv_b6d.hip)
    if (m->precision != CM_PRECISION_F16 && Ci_ref == Ci_pad && a.C0 % 16 == 0 && a.C1 % 16 == 0 && b6d_wanted(s.Co, s.out->Z, s.out->Y, s.out->X) &&
        cm::conv_b6d_pick(s.out->Z, s.out->Y, s.out->X, &op.b6d_bz, &op.b6d_by, &op.b6d_bx, &op.b6d_nw, &op.b6d_mbw)) {
      if (upload(m, pack_b6d(wi.data(), s.Co, Ci_ref, 27, cm::conv_b6d_nb(s.Co)), &op.d_wb6d)) return 1;
      op.b6d = true;
    }
  }
  // fp32 plan: the DownSample conv (3x3x3, stride 2, raw input: layers.py:91-97) on the direct six-term kernel -- one staging pass
  // and the split per staged element, no K split / second pass at quarter resolution.  OPT-IN (CM_DIAG=1 CM_B6S2=1): measured
  // SLOWER than the generic kernel on the ATC grid (32.2 vs 31.5 us at half resolution, 46.7 vs 18.7 + 7.0 us at quarter
  // resolution; step 1.402 vs 1.380 ms): with one row block per wave the kernel waits on its weight fragments (3 taps of
  // read-ahead = 576 matrix cycles, less than the L2 latency under load) and an 8-voxel halo per output voxel is staged per chunk
  if (s.ntaps == 27 && s.stride == 2 && !s.ups && !parity && !s.gn && !s.temb && !s.resid && !s.skip0 && m->precision != CM_PRECISION_F16 &&
      Ci_ref == Ci_pad && a.C0 % 16 == 0 && a.C1 % 16 == 0 && s.Co % 32 == 0 && s.Co == s.out->C && cm::diag_env("CM_B6S2") &&
      cm::conv_b6d_pick(s.out->Z, s.out->Y, s.out->X, &op.b6d_bz, &op.b6d_by, &op.b6d_bx, &op.b6d_nw, &op.b6d_mbw, 2, s.s0->Z, s.s0->Y, s.s0->X)) {
    if (const char *t = cm::diag_env("CM_B6S2_TILE")) {      // "bz,by,bx,nw": A/B of tile shapes
      int z = 0, y = 0, x = 0, w = 0;
      if (sscanf(t, "%d,%d,%d,%d", &z, &y, &x, &w) == 4 && s.out->Z % z == 0 && s.out->Y % y == 0 && s.out->X % x == 0 && z * y * x <= 32 * w) {
        op.b6d_bz = z; op.b6d_by = y; op.b6d_bx = x; op.b6d_nw = w; op.b6d_mbw = 1;
      }
    }
    if (upload(m, pack_b6d(wi.data(), s.Co, Ci_ref, 27, cm::conv_b6d_nb(s.Co)), &op.d_wb6d)) return 1;
    op.b6s2 = true;
  }
  // the UNet's first conv (C <= 8 data channels -> base): dedicated kernel, whole weight set in registers
  if (s.s0 == m->x8_act && s.ntaps == 27 && s.stride == 1 && !s.ups && !s.gn && !s.temb && !s.resid && !s.s1 &&
      s.Co % 32 == 0 && Ci_ref <= 8 && !cm::diag_env("CM_NO_FIRSTK")) {
    op.first_k = true;
    op.first_cin = Ci_ref <= 4 ? 4 : 8;
    const int cin = op.first_cin, NS = 27 * cin / 2, hc = cin / 2, ntn = s.Co / 32;
    std::vector<float> wp((size_t)ntn * NS * 64, 0.f);
    for (int nt = 0; nt < ntn; ++nt)
      for (int t = 0; t < 27; ++t)
        for (int pp = 0; pp < hc; ++pp)
          for (int lane = 0; lane < 64; ++lane) {
            const int co = nt * 32 + (lane & 31), ci = hc * (lane >> 5) + pp;   // MFMA (t, pp) contracts k = {hc*hh + pp}
            if (ci < Ci_ref) wp[((size_t)nt * NS + t * hc + pp) * 64 + lane] = wi[((size_t)co * Ci_ref + ci) * 27 + t];
          }
    if (upload(m, wp, &op.d_wfirst)) return 1;
  }
  float *dw = nullptr, *db = nullptr;
  if (upload(m, wf, &dw)) return 1;
  const int TN = 32 * op.NB, co_pad = (s.Co + TN - 1) / TN * TN;
  std::vector<float> bp((size_t)co_pad, 0.f);
  std::copy(b.host.begin(), b.host.end(), bp.begin());
  if (upload(m, bp, &db)) return 1;
  a.wfrag = dw; a.bias = db;
  if (s.stats && !cm::diag_env("CM_NO_FUSED_STATS")) op.stat_act = s.out;
  op.out_act = s.out;
  op.resid_act = s.resid;
  op.pm_off = s.pm_off;
  op.in0 = s.s0; op.in1 = s.s1;
  op.wname = s.wname; op.bname = s.bname;
  op.ref_taps = s.ntaps;
  op.temb_off = s.temb ? (int)(s.temb - m->temb_table) : -1;
  if (s.gn)
    for (int i = (int)m->ops.size() - 1; i >= 0; --i)
      if (m->ops[i].kind == OP_GNFIN && m->ops[i].gn_out == s.gn) { op.gn_op = i; break; }
  const bool h2_plan = m->precision == CM_PRECISION_F32 && !cm::diag_env("CM_NO_H2") && s.gn && op.gn_op >= 0 && s.silu;
  if (op.fin && h2_plan && h2_act_bounded(m, m->ops[op.gn_op], 1.0)) {
    const float ws = h2_wscale(w.host.data(), w.host.size());
    if (ws > 0.f) {
      if (dev_alloc(m, (void **)&op.d_wfin_h2, cm::CM_FIN_W_FLOATS * sizeof(float))) return 1;
      CM_HIP(cm::launch_fin_pack(op.d_wfin_src, op.d_wfin_h2, s.Co, 2, m->stream, ws));
      CM_HIP(hipStreamSynchronize(m->stream));
      op.h2_oscale = 1.f / ws;
    }
  }
  if (op.wino && op.d_wwino_b6 && h2_plan && h2_act_bounded(m, m->ops[op.gn_op], 4.0)) {
    std::vector<float> wh2;
    const float ws = h2_pack_wino(w.host.data(), s.Co, Ci_ref, Ci_pad, &wh2);
    if (ws > 0.f) {
      if (upload(m, wh2, &op.d_wwino_h2)) return 1;
      op.h2_oscale = 1.f / ws;
    }
  }
  // Tiny-spatial layers (one 54-voxel tile per sample at quarter resolution): the only way to
  // more parallelism AND less weight traffic per workgroup is to split K over workgroups;
  // a second pass sums the partials in a fixed order and applies the epilogue.
  const int nchunks = a.nch0 + a.nch1;
  static const int qr_vmax = cm::diag_env("CM_QR_VMAX") ? atoi(cm::diag_env("CM_QR_VMAX")) : 96;   // HERMES-CR-120 quarter resolution: 84 voxels
  if (s.ntaps == 27 && !parity && a.CK == 32 && nchunks >= 2 && s.out->V() <= qr_vmax && s.Co <= 256 && s.Co == s.out->C &&
      op.stat_act && !cm::diag_env("CM_NO_KSPLIT")) {
    static const int qr_ks = cm::diag_env("CM_QR_KS") ? atoi(cm::diag_env("CM_QR_KS")) : 4;
    op.ks = std::max(1, std::min(nchunks, s.out->V() > 64 ? std::min(qr_ks, 2) : qr_ks));   // (84 voxels: ks 2 3.06 ms, ks 4 3.12, none 3.16 per CR-120 step)
    const size_t need = (size_t)op.ks * m->cfg.max_batch * s.out->V() * s.Co;
    m->ks_scratch_floats = std::max(m->ks_scratch_floats, need);
    std::vector<float> zb((size_t)co_pad, 0.f);
    if (upload(m, zb, &op.d_zero_bias)) return 1;
  }
  // fuse the 1x1x1 skip conv when this conv runs on the 27-tap register-ring path without K split
  if (s.skip0 && s.ntaps == 27 && !parity && a.CK == 32 && !op.small_n && (!op.wino || op.NB == 1) && s.stride == 1 &&
      s.skip0->C % 32 == 0 && (!s.skip1 || s.skip1->C % 32 == 0) && !cm::diag_env("CM_NO_FUSE_SKIP")) {
    const Param &w2 = P(m, s.skip_w);
    const Param &b2 = P(m, s.skip_b);
    const int Ci2 = (int)w2.shape[1];
    if (Ci2 == s.skip0->C + (s.skip1 ? s.skip1->C : 0) && (int)w2.shape[0] == s.Co) {
      std::vector<float> w2f = pack_conv_weights(w2.host.data(), s.Co, Ci2, 1, Ci2, 32, op.NB);
      if (upload(m, w2f, &op.d_s2w)) return 1;
      std::vector<float> bf((size_t)co_pad, 0.f);
      for (int i = 0; i < s.Co; ++i) bf[i] = b.host[i] + b2.host[i];
      if (upload(m, bf, &op.d_bias_fused)) return 1;
      op.skip0 = s.skip0; op.skip1 = s.skip1; op.skip_w = s.skip_w; op.skip_b = s.skip_b;
    }
  }
  if (op.f16d && op.d_s2w) {
    const Param &w2 = P(m, s.skip_w);
    const int Cs = (int)w2.shape[1];
    if (Cs % 16 == 0 && s.skip0->C % 16 == 0 && (!s.skip1 || s.skip1->C % 16 == 0)) {
      if (upload(m, pack_f16d(w2.host.data(), s.Co, Cs, 1, s.Co % 64 == 0 ? 2 : 1), &op.d_w16d_skip)) return 1;
    } else {
      op.f16d = false;
    }
  }
  if (op.b6d && op.d_s2w) {
    const Param &w2 = P(m, s.skip_w);
    const int Cs = (int)w2.shape[1];
    if (Cs % 16 == 0 && s.skip0->C % 16 == 0 && (!s.skip1 || s.skip1->C % 16 == 0)) {
      if (upload(m, pack_b6d(w2.host.data(), s.Co, Cs, 1, cm::conv_b6d_nb(s.Co)), &op.d_wb6d_skip)) return 1;
    } else {
      op.b6d = false;
    }
  }
  // Lowest resolution (two z planes, <= 64 voxels per plane): whole-sample kernel with the GroupNorm finalisation of its
  // input inside (cm_conv_qr.hip), inference plan.  The K-split / generic set-up above stays for the training forward.
  if (s.ntaps == 27 && !parity && s.stride == 1 && !s.ups && s.gn && op.gn_op >= 0 && s.ci_valid < 0 && s.out->Z == 2 && s.s0->Z == 2 &&
      s.out->Y * s.out->X <= 64 && s.Co % 32 == 0 && s.Co == s.out->C && op.stat_act && (!s.skip0 || op.d_s2w) &&
      !cm::diag_env("CM_NO_QR")) {
    cm::QrArgs q{};
    q.C0 = a.C0; q.C1 = a.C1; q.Co = s.Co; q.Y = s.out->Y; q.X = s.out->X; q.groups = GN_GROUPS; q.gamma = m->ops[op.gn_op].gamma;
    if (op.d_s2w) { q.s2w = op.d_s2w; q.s2C0 = s.skip0->C; q.s2C1 = s.skip1 ? s.skip1->C : 0; }
    if (cm::conv_qr_ok(q)) {
      {
        const std::vector<float> wq = pack_qr(wi, s.Co, Ci_ref);
        if (upload(m, wq, &op.d_wqr)) return 1;
        op.wqr_floats = (long long)wq.size();
        if (m->precision != CM_PRECISION_F16 && Ci_ref % 64 == 0 && !cm::diag_env("CM_NO_QR_B6") && upload(m, pack_qr_b6(wq, s.Co, Ci_ref), &op.d_wqr_b6))
          return 1;
        if (op.d_wqr_b6 && h2_plan && h2_act_bounded(m, m->ops[op.gn_op], 1.0)) {
          std::vector<float> wh2;
          const float ws = h2_pack_qr(w.host.data(), s.Co, Ci_ref, &wh2);
          if (ws > 0.f) {
            if (upload(m, wh2, &op.d_wqr_h2)) return 1;
            op.h2_oscale = 1.f / ws;
          }
        }
      }
      if (op.d_s2w) {
        const Param &w2 = P(m, s.skip_w);
        if (upload(m, pack_qr_skip(w2.host.data(), s.Co, (int)w2.shape[1]), &op.d_wqr_skip)) return 1;
      }
      op.qr = true;
      m->ops[op.gn_op].qr_consumer = true;
    }
  }
  op.flops_per_sample = 2.0 * s.out->V() * s.Co * (double)Ci_ref * s.ntaps;
  op.label = s.wname;
  m->ops.push_back(op);
  return 0;
}

void add_stats(cm_model *m, const Act *a) {
  if (!cm::diag_env("CM_NO_FUSED_STATS")) return;  // the producing conv writes the statistics in its epilogue
  Op op;
  op.kind = OP_STATS; op.cls = K_NORM; op.act = a; op.label = "stats(" + a->name + ")";
  m->ops.push_back(op);
}

int add_gnfin(cm_model *m, const Act *g0, const Act *g1, const std::string &wname, const std::string &bname, float **gn_out) {
  Op op;
  op.kind = OP_GNFIN; op.cls = K_NORM; op.g0 = g0; op.g1 = g1; op.label = "gn_finalize(" + wname + ")";
  op.gname = wname; op.bename = bname;
  float *dg = nullptr, *db = nullptr;
  if (upload(m, P(m, wname).host, &dg)) return 1;
  if (upload(m, P(m, bname).host, &db)) return 1;
  op.gamma = dg; op.beta = db;
  const int Ct = g0->C + (g1 ? g1->C : 0);
  if (dev_alloc(m, (void **)&op.gn_out, (size_t)m->cfg.max_batch * 2 * Ct * sizeof(float))) return 1;
  *gn_out = op.gn_out;
  m->ops.push_back(op);
  return 0;
}

int build_ops(cm_model *m) {
  const cm_unet_config &c = m->cfg;
  const int L = m->L();
  int rc = 0;
  // spatial dims per level: conv k3 s2 p1 -> floor((n-1)/2)+1  (layers.py:84)
  std::vector<int> Zl{L}, Yl{c.rows}, Xl{c.cols};
  for (int l = 1; l < c.n_levels; ++l) {
    Zl.push_back((Zl.back() - 1) / 2 + 1);
    Yl.push_back((Yl.back() - 1) / 2 + 1);
    Xl.push_back((Xl.back() - 1) / 2 + 1);
  }
  for (int l = 1; l < c.n_levels; ++l)
    if (Zl[l] * 2 != Zl[l - 1] || Yl[l] * 2 != Yl[l - 1] || Xl[l] * 2 != Xl[l - 1])
      return fail("grid %dx%dx%d is not divisible by 2^%d: the reference's skip concat (unet.py:160) would fail too",
                  c.rows, c.cols, L, c.n_levels - 1);

  // time-embedding projection table offsets (one slice of nproj per res block)
  std::map<std::string, int> temb_off;
  {
    int off = 0;
    auto visit = [&](const BlockDesc &b) { if (b.kind == 0) { temb_off[b.prefix] = off; off += b.cout; } };
    for (auto &b : m->enc) visit(b);
    for (auto &b : m->bott) visit(b);
    for (auto &b : m->dec) visit(b);
    m->nproj = off;
  }
  if (dev_alloc(m, (void **)&m->temb_table, (size_t)TIME_ROWS * m->nproj * sizeof(float))) return 1;

  // input / output tensors
  auto xin = std::make_unique<Act>();
  xin->name = "input"; xin->C = 8; xin->Z = L; xin->Y = c.rows; xin->X = c.cols;
  if (dev_alloc(m, (void **)&xin->d, (size_t)c.max_batch * xin->V() * 8 * sizeof(float))) return 1;
  CM_HIP(hipMemset(xin->d, 0, (size_t)c.max_batch * xin->V() * 8 * sizeof(float)));
  m->x8 = xin->d; m->x8_act = xin.get();
  m->acts.push_back(std::move(xin));

  auto res_block = [&](const BlockDesc &b, const Act *x0, const Act *x1, Act **result) -> int {
    const int l = b.level;
    const std::string &p = b.prefix;
    float *gn1 = nullptr, *gn2 = nullptr;
    if (add_gnfin(m, x0, x1, p + ".normalize_1.weight", p + ".normalize_1.bias", &gn1)) return 1;
    Act *h1 = new_act(m, p + ".conv_1", b.cout, Zl[l], Yl[l], Xl[l], true, &rc);
    if (rc) return 1;
    ConvSpec c1; c1.s0 = x0; c1.s1 = x1; c1.gn = gn1; c1.silu = 1; c1.wname = p + ".conv_1.weight"; c1.bname = p + ".conv_1.bias";
    c1.temb = m->temb_table + temb_off[p]; c1.out = h1; c1.Co = b.cout; c1.stats = true;
    if (add_conv(m, c1)) return 1;
    add_stats(m, h1);
    if (add_gnfin(m, h1, nullptr, p + ".normalize_2.weight", p + ".normalize_2.bias", &gn2)) return 1;
    const Act *resid = x0;
    int mi_index = -1;
    if (b.cin != b.cout) {
      Act *r = new_act(m, p + ".match_input", b.cout, Zl[l], Yl[l], Xl[l], false, &rc);
      if (rc) return 1;
      ConvSpec cs; cs.s0 = x0; cs.s1 = x1; cs.ntaps = 1; cs.wname = p + ".match_input.weight"; cs.bname = p + ".match_input.bias";
      cs.out = r; cs.Co = b.cout;
      if (add_conv(m, cs)) return 1;
      mi_index = (int)m->ops.size() - 1;
      resid = r;
    } else if (x1) {
      return fail("identity skip with concatenated input is not expressible (block %s)", p.c_str());
    }
    Act *h2 = new_act(m, b.attention ? p + ".conv_2+skip" : p, b.cout, Zl[l], Yl[l], Xl[l], true, &rc);
    if (rc) return 1;
    ConvSpec c2; c2.s0 = h1; c2.gn = gn2; c2.silu = 1; c2.wname = p + ".conv_2.weight"; c2.bname = p + ".conv_2.bias";
    c2.resid = resid; c2.out = h2; c2.Co = b.cout; c2.stats = true; c2.pm_off = temb_off[p];
    if (mi_index >= 0) { c2.skip0 = x0; c2.skip1 = x1; c2.skip_w = p + ".match_input.weight"; c2.skip_b = p + ".match_input.bias"; }
    if (add_conv(m, c2)) return 1;
    if (mi_index >= 0 && m->ops.back().d_s2w) m->ops[mi_index].skip_if_fused = true;
    add_stats(m, h2);
    *result = h2;
    if (b.attention) {
      const std::string ap = p + ".attention";
      float *gna = nullptr;
      if (add_gnfin(m, h2, nullptr, ap + ".group_norm.weight", ap + ".group_norm.bias", &gna)) return 1;
      Act *qkv = new_act(m, ap + ".qkv", 3 * b.cout, Zl[l], Yl[l], Xl[l], false, &rc);
      if (rc) return 1;
      ConvSpec cq; cq.s0 = h2; cq.gn = gna; cq.silu = 0; cq.ntaps = 1; cq.wname = ap + ".mhsa.in_proj_weight"; cq.bname = ap + ".mhsa.in_proj_bias";
      cq.out = qkv; cq.Co = 3 * b.cout;
      if (add_conv(m, cq)) return 1;
      Act *ao = new_act(m, ap + ".core", b.cout, Zl[l], Yl[l], Xl[l], false, &rc);
      if (rc) return 1;
      Op at; at.kind = OP_ATTN; at.cls = K_ATTN; at.label = ap + ".core"; at.qkv = qkv->d; at.aout = ao->d; at.S = qkv->V(); at.E = b.cout;
      if ((size_t)2 * at.S * (at.E / ATTN_HEADS) * 4 > 160 * 1024)
        return fail("attention with %d tokens exceeds the LDS-resident K/V design", at.S);
      m->ops.push_back(at);
      Act *h3 = new_act(m, p, b.cout, Zl[l], Yl[l], Xl[l], true, &rc);
      if (rc) return 1;
      ConvSpec co; co.s0 = ao; co.ntaps = 1; co.wname = ap + ".mhsa.out_proj.weight"; co.bname = ap + ".mhsa.out_proj.bias";
      co.resid = h2; co.out = h3; co.Co = b.cout; co.stats = true;
      if (add_conv(m, co)) return 1;
      add_stats(m, h3);
      *result = h3;
      // Inference plan: the whole AttentionBlock as one launch per (head, sample) + the head combine
      // (cm_attn_block.hip) when the sample's tokens fit the LDS-resident design; the four generic ops above
      // stay in the list for the training forward (their activations feed the backward pass) and as fallback.
      const int nops = (int)m->ops.size();
      if (cm::attn_block_ok(qkv->V(), b.cout, ATTN_HEADS, GN_GROUPS) && !cm::diag_env("CM_NO_FUSED_ATTN") && !cm::diag_env("CM_NO_FUSED_STATS")) {
        Op fb;
        fb.kind = OP_ATTNBLK; fb.cls = K_ATTN; fb.label = ap + " (fused block)";
        fb.ab_x = h2; fb.ab_out = h3; fb.S = qkv->V(); fb.E = b.cout;
        fb.ab_gn = nops - 4; fb.ab_qkv = nops - 3; fb.ab_outc = nops - 1;
        if (m->ops[fb.ab_gn].kind != OP_GNFIN || m->ops[fb.ab_qkv].kind != OP_CONV || m->ops[nops - 2].kind != OP_ATTN ||
            m->ops[fb.ab_outc].kind != OP_CONV)
          return fail("attention block plan out of order");
        fb.win_name = ap + ".mhsa.in_proj_weight"; fb.wout_name = ap + ".mhsa.out_proj.weight";
        if (upload(m, P(m, fb.win_name).host, &fb.d_win)) return 1;
        if (upload(m, P(m, fb.wout_name).host, &fb.d_wout)) return 1;
        for (int i = nops - 4; i < nops; ++i) m->ops[i].in_attn_block = true;
        const size_t need = (size_t)ATTN_HEADS * m->cfg.max_batch * fb.S * fb.E;
        m->ks_scratch_floats = std::max(m->ks_scratch_floats, need);
        m->ops.push_back(fb);
      }
    }
    return 0;
  };

  // first conv (unet.py:32,142)
  Act *h = new_act(m, "first", c.base_channels, Zl[0], Yl[0], Xl[0], true, &rc);
  if (rc) return 1;
  {
    ConvSpec cf; cf.s0 = m->x8_act; cf.wname = "first.weight"; cf.bname = "first.bias"; cf.out = h; cf.Co = c.base_channels; cf.ci_valid = c.in_channels; cf.stats = true;
    if (add_conv(m, cf)) return 1;
    add_stats(m, h);
  }
  std::vector<Act *> outs{h};
  for (auto &b : m->enc) {
    if (b.kind == 0) {
      Act *r = nullptr;
      if (res_block(b, h, nullptr, &r)) return 1;
      h = r;
    } else {
      Act *d = new_act(m, b.prefix, b.cout, Zl[b.level + 1], Yl[b.level + 1], Xl[b.level + 1], true, &rc);
      if (rc) return 1;
      ConvSpec cd; cd.s0 = h; cd.stride = 2; cd.wname = b.prefix + ".downsample.weight"; cd.bname = b.prefix + ".downsample.bias"; cd.out = d; cd.Co = b.cout; cd.stats = true;
      if (add_conv(m, cd)) return 1;
      add_stats(m, d);
      h = d;
    }
    outs.push_back(h);
  }
  for (auto &b : m->bott) {
    Act *r = nullptr;
    if (res_block(b, h, nullptr, &r)) return 1;
    h = r;
  }
  for (auto &b : m->dec) {
    if (b.kind == 0) {
      Act *skip = outs.back();
      outs.pop_back();
      Act *r = nullptr;
      if (res_block(b, h, skip, &r)) return 1;  // torch.cat([h, out], dim=1): h first (unet.py:160)
      h = r;
    } else {
      const int l = b.level - 1;
      Act *u = new_act(m, b.prefix, b.cout, Zl[l], Yl[l], Xl[l], true, &rc);
      if (rc) return 1;
      ConvSpec cu; cu.s0 = h; cu.ups = 1; cu.wname = b.prefix + ".upsample.1.weight"; cu.bname = b.prefix + ".upsample.1.bias"; cu.out = u; cu.Co = b.cout; cu.stats = true;
      if (add_conv(m, cu)) return 1;
      add_stats(m, u);
      h = u;
    }
  }
  // final: GN -> SiLU -> conv (unet.py:118-122,164)
  {
    float *gnf = nullptr;
    if (add_gnfin(m, h, nullptr, "final.0.weight", "final.0.bias", &gnf)) return 1;
    auto eo = std::make_unique<Act>();
    eo->name = "final"; eo->C = 8; eo->Z = L; eo->Y = c.rows; eo->X = c.cols;
    if (dev_alloc(m, (void **)&eo->d, (size_t)c.max_batch * eo->V() * 8 * sizeof(float))) return 1;
    CM_HIP(hipMemset(eo->d, 0, (size_t)c.max_batch * eo->V() * 8 * sizeof(float)));
    m->eps_cl = eo->d;
    Act *eop = eo.get();
    m->act_by_name["final"] = eop;
    m->acts.push_back(std::move(eo));
    ConvSpec cf; cf.s0 = h; cf.gn = gnf; cf.silu = 1; cf.wname = "final.2.weight"; cf.bname = "final.2.bias"; cf.out = eop; cf.Co = c.out_channels;
    if (add_conv(m, cf)) return 1;
  }
  if (m->ks_scratch_floats && dev_alloc(m, (void **)&m->ks_scratch, 4 * m->ks_scratch_floats * sizeof(float))) return 1;
  return 0;
}

// Time-embedding tables for all 1000 rows (embeddings.py:24-30 + layers.py:35,62).
// f16 ACTIVATIONS (round 4; reduced-precision plan = BASELINE configs[4]; the reference's autocast stores every conv output as
// fp16, ddpm.py:116-120): a tensor is stored as _Float16 when the kernel that produces it can write f16 and every kernel that reads
// it can read f16 -- the direct f16 conv (source, residual, fused skip source, output), the first conv (output), the f16 stage-once
// upsample conv (source, output) and the last conv (source).  That covers the full- and half-resolution tensors of a grid with >= 4
// planes at half resolution except the inputs of the two stride-2 convs; quarter-resolution tensors (2 % of the bytes) stay fp32.
// GroupNorm statistics come from the fp32 accumulators of the producer; a kernel that cannot honour a flagged tensor fails loudly
// (run_conv) instead of misreading it.  CM_NO_H16 under CM_DIAG=1 keeps fp32 storage (A/B runs).
int plan_h16(cm_model *m) {
  if (m->precision != CM_PRECISION_F16 || cm::diag_env("CM_NO_H16")) return 0;
  auto writes16 = [](const Op &o) { return o.kind == OP_CONV && o.ks <= 1 && ((o.wino && o.f16d) || o.first_k || (o.ups && o.d_wups16)); };
  for (auto &ap : m->acts) {
    Act *a = ap.get();
    bool ok = false;
    for (const Op &o : m->ops)
      if (o.kind == OP_CONV && o.out_act == a && !o.skip_if_fused) ok = writes16(o);
    if (!ok) continue;
    for (const Op &o : m->ops) {
      if (!ok) break;
      if (o.kind == OP_ATTNBLK && o.ab_x == a) ok = false;
      if (o.kind == OP_ATTN && (o.qkv == a->d || o.aout == a->d)) ok = false;
      if (o.kind != OP_CONV) continue;
      const bool f16d = o.wino && o.f16d && o.ks <= 1;
      if (o.skip_if_fused) continue;                            // the stand-alone skip conv: absorbed by the block's conv_2 at inference
      if (o.in0 == a || o.in1 == a) ok = ok && (f16d || (o.in0 == a && !o.in1 && o.ups && o.d_wups16 && o.ks <= 1) || (o.small_n && o.ks <= 1));
      if (o.resid_act == a) ok = ok && f16d;
      if (o.skip0 == a || o.skip1 == a) ok = ok && f16d && o.d_w16d_skip;
    }
    a->h16 = ok;
  }
  return 0;
}

// GroupNorm finalisations whose only consumer is a Winograd conv of the table-driven kernel: that kernel can merge the producers'
// slot partials itself when they are few (<= 32 per sample: the half- and quarter-resolution layers) -- decided per launch in
// run_ops / run_conv, here only the structural part.  (conv_qr2 has finalised its own input this way since round 3.)
int plan_slot_consumers(cm_model *m) {
  if (cm::diag_env("CM_NO_SLOT_GN")) return 0;
  const int nops = (int)m->ops.size();
  for (int i = 0; i < nops; ++i) {
    Op &g = m->ops[i];
    if (g.kind != OP_GNFIN || g.qr_consumer || g.in_attn_block || g.atomic) continue;
    int ncons = 0, cons = -1;
    for (int j = 0; j < nops; ++j)
      if (m->ops[j].kind == OP_CONV && m->ops[j].gn_op == i) { ++ncons; cons = j; }
    if (ncons != 1) continue;
    const Op &c = m->ops[cons];
    g.from_slots = c.wino && !c.qr && c.ks <= 1 && !c.f16d && !c.b6d && !c.skip_if_fused && (c.d_wwino_b6 || c.d_wwino16);
  }
  return 0;
}

// Which GroupNorm finalisations of the INFERENCE plan can go without their launch (round 4): the source tensors' producers add
// exact fixed-point sums to accumulator rows (cm_stat_atomic), the consuming conv finalises them in its prologue (six-term Winograd
// kernel) or through a fall-back launch that costs what gn_finalize cost.  Eligible: fp32 plan; every source tensor produced by an
// ordinary conv launch (no K split, not the whole-sample quarter-resolution kernel -- those feed slot partials to consumers that
// merge slots); no consumer that merges slots itself (conv_qr).  A tensor takes accumulators only if ALL GroupNorms over it do.
// MEASURED SLOWER than the gn_finalize launches it removes (round 4, same-box: 1.456 vs 1.413 ms per step; removing the 13 launches
// outright would be worth 88 us): per full-resolution layer the 663 k 8-byte atomic adds -- 108 per accumulator word -- cost 14-17 us
// (conv_first 26 -> 40 us, the stage-once upsample conv 86 -> 118) and the consumer prologue ~3 us, against the 6.4 us launch.
// Opt-in only: CM_DIAG=1 CM_ASTAT=1 (kept with its parity test as the record of the experiment).
int plan_astat(cm_model *m) {
  if (m->precision == CM_PRECISION_F16 || !cm::diag_env("CM_ASTAT")) return 0;
  const int nops = (int)m->ops.size();
  std::map<const Act *, int> producer;
  for (int i = 0; i < nops; ++i)
    if (m->ops[i].kind == OP_CONV && m->ops[i].stat_act) producer[m->ops[i].stat_act] = i;
  std::vector<char> cand((size_t)nops, 0);
  auto act_ok = [&](const Act *a) {
    auto it = producer.find(a);
    if (it == producer.end()) return false;
    const Op &po = m->ops[it->second];
    return po.ks <= 1 && !po.qr && !po.skip_if_fused && a->part != nullptr;
  };
  for (int i = 0; i < nops; ++i) {
    Op &g = m->ops[i];
    if (g.kind != OP_GNFIN || g.qr_consumer || g.in_attn_block) continue;
    bool ok = act_ok(g.g0) && (!g.g1 || act_ok(g.g1));
    int ncons = 0;
    for (int j = 0; j < nops && ok; ++j)
      if (m->ops[j].kind == OP_CONV && m->ops[j].gn_op == i) { ++ncons; if (m->ops[j].qr) ok = false; }
    cand[(size_t)i] = ok && ncons > 0;
  }
  for (bool changed = true; changed;) {            // a tensor keeps accumulators only if every GroupNorm over it is a candidate
    changed = false;
    for (int i = 0; i < nops; ++i) {
      if (!cand[(size_t)i]) continue;
      const Op &g = m->ops[i];
      for (int j = 0; j < nops; ++j) {
        const Op &h = m->ops[j];
        if (h.kind != OP_GNFIN || cand[(size_t)j]) continue;
        if (h.g0 == g.g0 || h.g0 == g.g1 || (h.g1 && (h.g1 == g.g0 || h.g1 == g.g1))) { cand[(size_t)i] = 0; changed = true; break; }
      }
    }
  }
  int totC = 0;
  for (int i = 0; i < nops; ++i) {
    if (!cand[(size_t)i]) continue;
    Op &g = m->ops[i];
    g.atomic = true;
    for (const Act *a : {g.g0, g.g1})
      if (a && a->aoff < 0) { const_cast<Act *>(a)->aoff = totC; totC += a->C; }
  }
  if (!totC) return 0;
  m->astat_C = totC;
  const size_t bytes = (size_t)m->cfg.max_batch * totC * 3 * sizeof(unsigned long long);
  if (dev_alloc(m, (void **)&m->astat_all, bytes)) return 1;
  CM_HIP(hipMemset(m->astat_all, 0, bytes));
  return 0;
}

int build_time_table(cm_model *m) {
  const cm_unet_config &c = m->cfg;
  const int te = c.base_channels, tx = c.base_channels * c.time_multiple;
  std::vector<float> Wd((size_t)m->nproj * tx), bd((size_t)m->nproj);
  int off = 0;
  auto visit = [&](const BlockDesc &b) {
    if (b.kind != 0) return;
    const Param &w = P(m, b.prefix + ".dense_1.weight");
    const Param &bb = P(m, b.prefix + ".dense_1.bias");
    std::copy(w.host.begin(), w.host.end(), Wd.begin() + (size_t)off * tx);
    std::copy(bb.host.begin(), bb.host.end(), bd.begin() + off);
    off += b.cout;
  };
  for (auto &b : m->enc) visit(b);
  for (auto &b : m->bott) visit(b);
  for (auto &b : m->dec) visit(b);
  float *dT, *dW1, *db1, *dW2, *db2, *dWd, *dbd;
  if (upload(m, P(m, "time_embeddings.time_blocks.0.weight").host, &dT)) return 1;
  if (upload(m, P(m, "time_embeddings.time_blocks.1.weight").host, &dW1)) return 1;
  if (upload(m, P(m, "time_embeddings.time_blocks.1.bias").host, &db1)) return 1;
  if (upload(m, P(m, "time_embeddings.time_blocks.3.weight").host, &dW2)) return 1;
  if (upload(m, P(m, "time_embeddings.time_blocks.3.bias").host, &db2)) return 1;
  if (upload(m, Wd, &dWd)) return 1;
  if (upload(m, bd, &dbd)) return 1;
  m->d_time[0] = dT; m->d_time[1] = dW1; m->d_time[2] = db1; m->d_time[3] = dW2; m->d_time[4] = db2;
  m->d_time[5] = dWd; m->d_time[6] = dbd;
  CM_HIP(cm::launch_time_mlp(dT, dW1, db1, dW2, db2, dWd, dbd, te, tx, m->nproj, TIME_ROWS, nullptr, m->temb_table, nullptr, m->stream));
  CM_HIP(hipStreamSynchronize(m->stream));
  return 0;
}

// ------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------
// One convolution op of the plan for the `B` samples starting at `b0` (see run_ops).
int run_combine(cm_model *m, cm::CombineArgs &cb, hipStream_t st);
struct Op;
static thread_local const Op *tl_fin_next = nullptr;   // GroupNorm finalisation waiting to ride on the next K-split second pass
static thread_local int tl_fin_b0 = 0;
static thread_local bool tl_fin_done = false;

// the whole-sample kernel of the lowest resolution (inference plan): the GroupNorm statistics of its sources come raw
int run_conv_qr(cm_model *m, Op &op, int B, hipStream_t st, int b0) {
  const Op &gop = m->ops[op.gn_op];
  const Act *g0 = gop.g0, *g1 = gop.g1;
  const cm::ConvArgs &ca = op.ca;
  const size_t V = (size_t)ca.Zo * ca.Yo * ca.Xo;
  cm::QrArgs q{};
  q.src0 = ca.src0 + (size_t)b0 * V * ca.C0; q.C0 = ca.C0;
  q.src1 = ca.src1 ? ca.src1 + (size_t)b0 * V * ca.C1 : nullptr; q.C1 = ca.C1;
  q.part0 = g0->part + (size_t)b0 * g0->nslots * g0->C * 2; q.cnt0 = g0->cnt + (size_t)b0 * g0->nslots; q.ns0 = g0->nslots;
  if (g1) { q.part1 = g1->part + (size_t)b0 * g1->nslots * g1->C * 2; q.cnt1 = g1->cnt + (size_t)b0 * g1->nslots; q.ns1 = g1->nslots; }
  if (g0->C != ca.C0 || (g1 ? g1->C : 0) != ca.C1) return fail("quarter-resolution conv %s: statistics and sources disagree", op.label.c_str());
  q.gamma = gop.gamma; q.beta = gop.beta; q.groups = GN_GROUPS; q.eps = GN_EPS; q.silu = ca.silu;
  if (op.dbg_raw) { q.gamma = q.beta = nullptr; q.raw = 1; q.silu = 0; }
  q.wq = op.d_wqr; q.bias = ca.bias;
  q.wq6 = op.d_wqr_b6;
  q.three = (m->precision == CM_PRECISION_F32R && !m->train_fwd) ? 1 : 0;
  // default plan, inference-only handle: the f16 two-way-split form (bounded input: GroupNorm + SiLU inside the kernel)
  if (m->precision == CM_PRECISION_F32 && !m->train_fwd && !m->h2_stale && !op.h2_off && op.d_wqr_h2 && (!op.dbg_raw || op.dbg_h2)) {
    q.wq6 = op.d_wqr_h2; q.three = 2; q.h2_oscale = op.h2_oscale;
  }
  q.temb = ca.temb; q.temb_stride = ca.temb_stride; q.tidx = ca.tidx + b0;
  q.resid = ca.resid ? ca.resid + (size_t)b0 * V * ca.res_cs : nullptr; q.res_cs = ca.res_cs;
  if (m->train_fwd) {
    // training forward (six-term form only, see run_conv): Dropout3d multipliers on the activated input, the time-embedding rows
    // of this batch, the block's skip conv as its own op (its output arrives as the residual)
    if (op.pm_off >= 0) { q.pm = m->dropmask + (size_t)b0 * m->nproj + op.pm_off; q.pm_stride = m->nproj; }
    if (m->use_train_temb && op.temb_off >= 0) { q.temb = m->train_temb + op.temb_off; q.tidx = m->train_iota + b0; }
  } else if (op.d_wqr_skip) {
    q.s2w = op.d_wqr_skip;
    q.s2src0 = op.skip0->d + (size_t)b0 * V * op.skip0->C; q.s2C0 = op.skip0->C;
    q.s2src1 = op.skip1 ? op.skip1->d + (size_t)b0 * V * op.skip1->C : nullptr; q.s2C1 = op.skip1 ? op.skip1->C : 0;
    q.resid = nullptr;
    q.bias = op.d_bias_fused;
  }
  q.out = ca.out + (size_t)b0 * V * ca.out_cs; q.out_cs = ca.out_cs; q.Co = ca.Co; q.B = B; q.Y = ca.Yo; q.X = ca.Xo;
  const int ns = ca.Yo * ca.Xo > 32 ? 4 : 2;
  q.stat_C = op.stat_act->C;
  q.stat_part = op.stat_act->part + (size_t)b0 * ns * q.stat_C * 2;
  q.stat_cnt = op.stat_act->cnt + (size_t)b0 * ns;
  op.stat_act->nslots = ns;
  op.prof_B = B;
  if (m->train_fwd && !cm::conv_qr2_b6_ok(q)) return fail("quarter-resolution conv %s: no six-term form for the training forward", op.label.c_str());
  CM_HIP(cm::launch_conv_qr(q, st));
  return 0;
}

int run_conv(cm_model *m, Op &op, int B, hipStream_t st, int b0, int slab) {
  if (op.skip_if_fused && !m->train_fwd) return 0;  // absorbed by the block's conv_2 (inference plan)
  // (training forward: the same whole-sample kernel in its six-term form -- it finalises the GroupNorm of its input from the
  //  producers' partials like the inference plan does; the gn_finalize op still runs there for the backward's mean / rstd rows)
  static const bool no_train_qr = cm::diag_env("CM_NO_TRAIN_QR") != nullptr;
  if (op.qr && (!m->train_fwd || (op.d_wqr_b6 && op.train_qr && !no_train_qr))) return run_conv_qr(m, op, B, st, b0);
  if (op.tuned_B < 0 && op.wino) {
    // feasibility is known only with the launch geometry: a grid the Winograd tiles do not fit falls back to the
    // direct kernel (its fragments `wfrag` are packed for every conv; NB = 1 was fixed before packing)
    cm::ConvArgs a = op.ca;
    a.bs = 1;
    bool ok = cm::conv_wino_pick(a.Zo, a.Yo, a.Xo, &a.bz, &a.by, &a.bx);
    if (ok) {
      a.ntz = a.Zo / a.bz; a.nty = (a.Yo + a.by - 1) / a.by; a.ntx = (a.Xo + a.bx - 1) / a.bx;
      ok = cm::conv_wino_ok(a);
    }
    if (ok) {
      op.ca = a;
      op.MB = 4;                     // statistics slots per tile: the four (a, b) output sub-blocks
      op.tuned_B = B;
    } else {
      op.wino = false;
    }
  }
  if (op.tuned_B < 0 && op.first_k) {
    const cm::ConvArgs keep = op.ca;
    pick_tile_first(op);
    if (cm::conv_first_ok(op.ca, op.first_cin)) {
      op.tuned_B = B;
    } else {
      op.ca = keep;                  // e.g. more columns than the full-X tile holds: the direct kernel takes it
      op.first_k = false;
    }
  }
  if (op.tuned_B < 0) {
    pick_tile(op, TUNE_BATCH);
    op.tuned_B = B;
    std::vector<int> hv((size_t)cm::conv_halo_voxels(op.ca)), mt((size_t)32 * op.MB);
    cm::conv_build_tables(op.ca, op.MB, hv.data(), mt.data());
    if (!op.d_hvtab) {
      if (dev_alloc(m, (void **)&op.d_hvtab, 16384 * sizeof(int))) return 1;
      if (dev_alloc(m, (void **)&op.d_mtab, 256 * sizeof(int))) return 1;
    }
    if (hv.size() > 16384) return fail("halo box too large");
    CM_HIP(hipMemcpyAsync(op.d_hvtab, hv.data(), hv.size() * sizeof(int), hipMemcpyHostToDevice, st));
    CM_HIP(hipMemcpyAsync(op.d_mtab, mt.data(), mt.size() * sizeof(int), hipMemcpyHostToDevice, st));
    CM_HIP(hipStreamSynchronize(st));
    op.ca.hvtab = op.d_hvtab;
    op.ca.mtab = op.d_mtab;
  }
  cm::ConvArgs ca = op.ca;
  ca.B = B;
  op.prof_B = B;
  ca.nts = (B + ca.bs - 1) / ca.bs;
  const size_t Vs = (size_t)ca.Zs * ca.Ys * ca.Xs, Vo = (size_t)ca.Zo * ca.Yo * ca.Xo;
  // f16 tensors of the reduced-precision plan (plan_h16): half the bytes per element, so half the float offset; the mask tells the
  // kernel which of its tensors they are (the training forward never sees them: training refuses f16 handles)
  auto is16 = [&](const Act *t) { return t && t->h16 && !m->train_fwd; };
  auto adv = [&](const float *p, size_t elems, bool h) { return p + (h ? elems / 2 : elems); };
  ca.h16 = (is16(op.in0) ? 1 : 0) | (is16(op.in1) ? 2 : 0) | (is16(op.out_act) ? 4 : 0) | (is16(op.resid_act) ? 8 : 0);
  ca.src0 = adv(ca.src0, (size_t)b0 * Vs * ca.C0, is16(op.in0));
  if (ca.src1) ca.src1 = adv(ca.src1, (size_t)b0 * Vs * ca.C1, is16(op.in1));
  if (ca.gn) ca.gn += (size_t)b0 * 2 * (ca.C0 + ca.C1);
  ca.tidx += b0;
  if (m->train_fwd && op.pm_off >= 0) {
    ca.pm = m->dropmask + (size_t)b0 * m->nproj + op.pm_off;
    ca.pm_stride = m->nproj;
  }
  if (m->use_train_temb && op.temb_off >= 0) {
    ca.temb = m->train_temb + op.temb_off;
    ca.tidx = m->train_iota + b0;
  }
  if (ca.resid) ca.resid = adv(ca.resid, (size_t)b0 * Vo * ca.res_cs, is16(op.resid_act));
  if (op.d_s2w && !m->train_fwd) {
    if (!op.wino && 32 * op.MB > cm::conv_halo_voxels(ca)) return fail("fused skip conv: tile rows exceed the staged box");
    ca.s2w = op.d_s2w;
    ca.s2src0 = adv(op.skip0->d, (size_t)b0 * Vo * op.skip0->C, is16(op.skip0)); ca.s2C0 = op.skip0->C;
    ca.s2src1 = op.skip1 ? adv(op.skip1->d, (size_t)b0 * Vo * op.skip1->C, is16(op.skip1)) : nullptr; ca.s2C1 = op.skip1 ? op.skip1->C : 0;
    ca.h16 = (ca.h16 & ~8) | (is16(op.skip0) ? 16 : 0) | (is16(op.skip1) ? 32 : 0);
    ca.resid = nullptr;
    ca.bias = op.d_bias_fused;
  }
  ca.out = const_cast<float *>(adv(ca.out, (size_t)b0 * Vo * ca.out_cs, is16(op.out_act)));
  int ns = 0;
  if (op.stat_act) {
    ns = ca.ntz * ca.nty * ca.ntx * op.MB * (ca.par ? 8 : 1);
    if (ns > MAX_SLOTS) return fail("statistics slots %d exceed %d", ns, MAX_SLOTS);
    ca.stat_C = op.stat_act->C;
    ca.stat_ns = ns;
    ca.stat_part = op.stat_act->part + (size_t)b0 * ns * ca.stat_C * 2;
    ca.stat_cnt = op.stat_act->cnt + (size_t)b0 * ns;
  }
  // ---- which kernel will run (needed up front: only some kernels speak the accumulator statistics of round 4) -------------------
  static const bool no_train_b6 = cm::diag_env("CM_NO_TRAIN_B6") != nullptr;
  // relaxed fp32 plan (cm_model_set_precision): the six-term kernels issue only their three leading cross terms (inference forward)
  const bool relaxed = m->precision == CM_PRECISION_F32R && !m->train_fwd;
  // default plan: the f16 two-way-split form on layers with bounded input (inference-only handles, see Op::d_wfin_h2)
  const bool h2_live = m->precision == CM_PRECISION_F32 && !m->train_fwd && !m->h2_stale && !op.h2_off;
  cm::ConvArgs s2a = ca;
  s2a.bz = op.b6d_bz; s2a.by = op.b6d_by; s2a.bx = op.b6d_bx;
  // (the training forward as well: exact splits, fp32 accumulate; its fragments follow every optimizer step)
  const bool take_b6s2 = op.b6s2 && !(m->train_fwd && no_train_b6) && m->precision != CM_PRECISION_F16 && !ca.h16 && !ca.pm &&
                         cm::conv_b6d_ok(s2a, op.b6d_nw, op.b6d_mbw);
  const int ks = take_b6s2 ? 1 : op.ks;            // (the direct stride-2 kernel replaces the K split)
  const bool take_ups = ks <= 1 && op.ups && (op.d_wups16 || !(op.d_wfrag16 && !m->train_fwd));
  const bool take_f16d = ks <= 1 && !take_ups && op.wino && op.f16d && !m->train_fwd;
  const bool take_b6d = take_b6s2 || (ks <= 1 && !take_ups && !take_f16d && op.wino && op.b6d && !m->train_fwd && m->precision != CM_PRECISION_F16);
  const bool take_wino = ks <= 1 && !take_ups && !take_f16d && !take_b6d && op.wino;
  // The slot count of the output tensor is written ONCE per launch with the count of the kernel that runs: the batch lanes
  // enqueue from two host threads, and the other lane's gn_finalize reads it -- a generic count first and the upsample /
  // direct kernel's own count afterwards left a window in which that reader saw the wrong number of slots (a rare wrong
  // statistic: the one-off failure of the two-lane bit-identity test in round 4).  All lanes write the same value.
  if (op.stat_act && ks <= 1 && !take_ups && !take_f16d && !take_b6d) op.stat_act->nslots = ns;
  const bool wino_f16 = take_wino && op.d_wwino16 && !m->train_fwd;     // reduced-precision plan: f16 operands in the inference forward
  // two-tile layers / the full-resolution tile, fp32 plan: six-term bf16 products (the training forward as well: exact splits, fp32
  // accumulate; its fragments follow every optimizer step)
  const bool wino_b6 = take_wino && !wino_f16 && op.d_wwino_b6 && !(m->train_fwd && (no_train_b6 || m->precision == CM_PRECISION_F16)) &&
                       cm::conv_wino_b6_ok(ca.bz, ca.by, ca.bx, ca.Co, ca.Zo);
  if (ca.h16) {
    // only these kernels read / write f16 tensors; anything else here would misread them silently
    const bool ok16 = take_f16d || (take_ups && op.d_wups16 && !(ca.h16 & ~5)) || (op.first_k && !(ca.h16 & ~4) && ks <= 1 && !take_ups && !op.wino) ||
                      (op.small_n && !(ca.h16 & ~3) && ks <= 1 && !take_ups && !op.wino && !op.first_k);
    if (!ok16) return fail("conv %s: f16 tensors (mask %d) reach a kernel without f16 tensor support", op.label.c_str(), ca.h16);
  }
  // consumer side: the GroupNorm of the input comes from the producers' accumulator rows -- finalised inside the six-term
  // Winograd kernel, or by a fall-back launch that writes the rows gn_finalize would have written
  if (!m->train_fwd && m->astat_all && op.gn_op >= 0 && m->ops[op.gn_op].atomic && ca.gn) {
    const Op &g = m->ops[op.gn_op];
    ca.gs0 = m->astat_all + ((size_t)b0 * m->astat_C + g.g0->aoff) * 3;
    ca.gs1 = g.g1 ? m->astat_all + ((size_t)b0 * m->astat_C + g.g1->aoff) * 3 : nullptr;
    ca.gs_C = m->astat_C; ca.gs_gamma = g.gamma; ca.gs_beta = g.beta; ca.gs_groups = GN_GROUPS; ca.gs_eps = GN_EPS;
    if (wino_b6) {
      ca.gn = nullptr;
    } else {
      CM_HIP(cm::launch_gn_from_sums(ca, (int)Vs, const_cast<float *>(ca.gn), st));
      ca.gs0 = ca.gs1 = nullptr;
    }
  }
  // ... or from their slot partials when run_ops skipped the gn_finalize launch on that promise (few slots)
  if (!m->train_fwd && op.gn_op >= 0 && m->ops[op.gn_op].fin_skipped[slab & 3] && ca.gn) {
    const Op &g = m->ops[op.gn_op];
    const Act *g0 = g.g0, *g1 = g.g1;
    const int nbw = cm::conv_wino_nbw_run(ca, wino_f16);
    const bool p_kernel = take_wino && cm::conv_wino_two_step(ca.bz, ca.by, ca.bx, wino_f16, nbw) && (nbw == 2 || wino_b6);
    if (p_kernel) {
      ca.gp0 = g0->part + (size_t)b0 * g0->nslots * g0->C * 2; ca.gc0 = g0->cnt + (size_t)b0 * g0->nslots; ca.gns0 = g0->nslots;
      if (g1) { ca.gp1 = g1->part + (size_t)b0 * g1->nslots * g1->C * 2; ca.gc1 = g1->cnt + (size_t)b0 * g1->nslots; ca.gns1 = g1->nslots; }
      ca.gs_gamma = g.gamma; ca.gs_beta = g.beta; ca.gs_groups = GN_GROUPS; ca.gs_eps = GN_EPS;
      ca.gn = nullptr;
    } else {
      // (a kernel that cannot: the launch that was skipped, now)
      const int Ct = g0->C + (g1 ? g1->C : 0);
      CM_HIP(cm::launch_gn_finalize(g0->part + (size_t)b0 * g0->nslots * g0->C * 2, g0->cnt + (size_t)b0 * g0->nslots, g0->nslots, g0->C,
                                    g1 ? g1->part + (size_t)b0 * g1->nslots * g1->C * 2 : nullptr, g1 ? g1->cnt + (size_t)b0 * g1->nslots : nullptr,
                                    g1 ? g1->nslots : 0, g1 ? g1->C : 0, g0->V(), g.gamma, g.beta, GN_GROUPS, GN_EPS,
                                    g.gn_out + (size_t)b0 * 2 * Ct, nullptr, B, st));
    }
  }
  // producer side: this launch adds its output's statistics to the accumulator rows instead of writing slot partials
  const bool use_astat = !m->train_fwd && m->astat_all && op.stat_act && op.stat_act->aoff >= 0;
  auto to_astat = [&]() {
    if (!use_astat) return;
    ca.astat = m->astat_all + ((size_t)b0 * m->astat_C + op.stat_act->aoff) * 3;
    ca.astat_C = m->astat_C;
    ca.stat_part = nullptr; ca.stat_cnt = nullptr;
    m->astat_clean[slab & 3].B = 0;
  };
  if (use_astat && ks > 1) return fail("conv %s: a K-split layer cannot feed accumulator statistics", op.label.c_str());
  if (ks > 1) {
    cm::ConvArgs ka = ca;
    const int V = op.out_act->V();
    float *scratch = m->ks_scratch + (size_t)slab * m->ks_scratch_floats;
    ka.temb = nullptr; ka.resid = nullptr; ka.stat_part = nullptr; ka.bias = op.d_zero_bias;
    ka.out = scratch; ka.out_cs = ka.Co;
    ka.ks = op.ks; ka.kpart = (long long)B * V * ka.Co;
    CM_HIP(cm::launch_conv(ka, op.MB, op.NB, st));
    cm::CombineArgs cb{};
    cb.part = scratch; cb.S = op.ks; cb.stride = ka.kpart;
    cb.bias = ca.bias; cb.temb = ca.temb; cb.temb_stride = ca.temb_stride; cb.tidx = ca.tidx;
    cb.resid = ca.resid; cb.res_cs = ca.res_cs;
    cb.out = ca.out; cb.C = ka.Co; cb.V = V; cb.B = B;
    cb.nslots = (V + 31) / 32;
    cb.stat_part = op.stat_act->part + (size_t)b0 * cb.nslots * cb.C * 2;
    cb.stat_cnt = op.stat_act->cnt + (size_t)b0 * cb.nslots;
    op.stat_act->nslots = cb.nslots;
    if (run_combine(m, cb, st)) return 1;
  } else if (take_ups) {
    // upsample conv: stage-once parity kernel with its own source tile / statistics slots (f16 operands under the
    // reduced-precision plan's inference forward)
    if (op.d_wups16 && !m->train_fwd) { ca.wfrag = op.d_wups16; ca.wpar_stride = op.wups16_stride; ca.f16 = 1; }
    else if (op.d_wups_b6 && !(m->train_fwd && cm::diag_env("CM_NO_TRAIN_B6"))) { ca.wfrag = op.d_wups_b6; ca.wpar_stride = op.wups_b6_stride; ca.f16 = relaxed ? 3 : 2; }   // six-term bf16 products (training forward too); relaxed plan: three
    if (ca.f16 == 2 && h2_live && op.d_wups_h2 && op.in0 && op.in0->part && op.in0->nslots > 0 && !m->astat_all && (!op.dbg_raw || op.dbg_h2)) {
      // default plan, inference-only handle: h2 with the sample's block exponent from the source tensor's slot statistics
      const Act *si = op.in0;
      ca.wfrag = op.d_wups_h2; ca.f16 = 4; ca.h2_oscale = op.h2_oscale;
      ca.gp0 = si->part + (size_t)b0 * si->nslots * si->C * 2; ca.gc0 = si->cnt + (size_t)b0 * si->nslots; ca.gns0 = si->nslots;
    }
    ca.bz = op.ups_tz; ca.by = op.ups_ty; ca.bx = op.ups_tx;
    ca.ntz = ca.Zs / ca.bz; ca.nty = ca.Ys / ca.by; ca.ntx = ca.Xs / ca.bx;
    if (op.stat_act) {
      const int nsu = cm::conv_ups_slots(ca, op.ups_mbw);
      if (nsu > MAX_SLOTS) return fail("statistics slots %d exceed %d", nsu, MAX_SLOTS);
      ca.stat_ns = nsu;
      ca.stat_part = op.stat_act->part + (size_t)b0 * nsu * ca.stat_C * 2;
      ca.stat_cnt = op.stat_act->cnt + (size_t)b0 * nsu;
      op.stat_act->nslots = nsu;
    }
    to_astat();
    CM_HIP(cm::launch_conv_ups(ca, op.ups_mbw, op.ups_planes, op.NB, st));
  } else if (take_f16d) {
    // reduced-precision plan: direct f16 kernel with its own tile geometry / statistics slots
    ca.bz = op.f16d_bz; ca.by = op.f16d_by; ca.bx = op.f16d_bx;
    ca.wfrag = op.d_w16d;
    if (ca.s2w) ca.s2w = op.d_w16d_skip;
    if (op.stat_act) {
      const int ns16 = cm::conv_f16d_slots(ca, op.f16d_mbw);
      if (ns16 > MAX_SLOTS) return fail("statistics slots %d exceed %d", ns16, MAX_SLOTS);
      ca.stat_ns = ns16;
      ca.stat_part = op.stat_act->part + (size_t)b0 * ns16 * ca.stat_C * 2;
      ca.stat_cnt = op.stat_act->cnt + (size_t)b0 * ns16;
      op.stat_act->nslots = ns16;
    }
    CM_HIP(cm::launch_conv_f16d(ca, op.f16d_mbw, st));
    if (use_astat) {                               // (this kernel writes slot partials only: convert them)
      m->astat_clean[slab & 3].B = 0;
      CM_HIP(cm::launch_slots_to_sums(ca.stat_part, ca.stat_cnt, ca.stat_ns, ca.stat_C, B,
                                      m->astat_all + ((size_t)b0 * m->astat_C + op.stat_act->aoff) * 3, m->astat_C, st));
    }
  } else if (take_b6d) {
    // fp32 plan, inference forward: direct six-term kernel with its own tile geometry / statistics slots
    ca.bz = op.b6d_bz; ca.by = op.b6d_by; ca.bx = op.b6d_bx;
    ca.wfrag = op.d_wb6d;
    if (ca.s2w) ca.s2w = op.d_wb6d_skip;
    if (op.stat_act) {
      const int ns6 = cm::conv_b6d_slots(ca, op.b6d_nw, op.b6d_mbw);
      if (ns6 > MAX_SLOTS) return fail("statistics slots %d exceed %d", ns6, MAX_SLOTS);
      ca.stat_ns = ns6;
      ca.stat_part = op.stat_act->part + (size_t)b0 * ns6 * ca.stat_C * 2;
      ca.stat_cnt = op.stat_act->cnt + (size_t)b0 * ns6;
      op.stat_act->nslots = ns6;
    }
    to_astat();
    CM_HIP(cm::launch_conv_b6d(ca, op.b6d_nw, op.b6d_mbw, st));
  } else if (take_wino) {
    ca.wfrag = wino_f16 ? op.d_wwino16 : op.d_wwino;
    if (wino_b6) { ca.wfrag = op.d_wwino_b6; ca.f16 = relaxed ? 3 : 2; }
    if (wino_b6 && h2_live && op.d_wwino_h2 && (!op.dbg_raw || op.dbg_h2)) { ca.wfrag = op.d_wwino_h2; ca.f16 = 4; ca.h2_oscale = op.h2_oscale; }
    to_astat();
    CM_HIP(cm::launch_conv_wino(ca, wino_f16, st));
  } else if (op.first_k) {
    to_astat();
    CM_HIP(cm::launch_conv_first(ca, op.first_cin, op.d_wfirst, st));
  } else if (op.small_n) {
    bool done = false;
    if (op.fin) {
      cm::ConvArgs fa = ca;
      fa.by = op.fin_by; fa.bx = op.fin_bx;
      const bool f16 = op.d_wfin16 && !m->train_fwd;
      const bool h2 = !f16 && h2_live && op.d_wfin_h2 && (!op.dbg_raw || op.dbg_h2);
      fa.h2_oscale = op.h2_oscale;
      if (cm::conv_fin_ok(fa)) {
        CM_HIP(cm::launch_conv_fin(fa, f16 ? op.d_wfin16 : (h2 ? op.d_wfin_h2 : op.d_wfin), f16 ? 1 : (h2 ? 3 : (relaxed ? 2 : 0)), st));
        done = true;
      }
    }
    if (!done) CM_HIP(cm::launch_conv_smalln(ca, op.MB, op.d_wsmall, st));
  } else {
    if (op.d_w1x1_16 && !m->train_fwd && cm::conv1x1_f16_ok(ca, op.NB)) {
      ca.wfrag = op.d_w1x1_16;
      CM_HIP(cm::launch_conv1x1_f16(ca, op.NB, st));
      return 0;
    }
    if (op.d_wfrag16 && !m->train_fwd && cm::conv_par_f16_variant(op.MB, op.NB, ca.bz, ca.by, ca.bx)) {
      ca.wfrag = op.d_wfrag16; ca.wpar_stride = op.wpar_stride16; ca.f16 = 1;   // f16 operands, fp32 accumulate
    }
    to_astat();
    CM_HIP(cm::launch_conv(ca, op.MB, op.NB, st));
  }
  return 0;
}

// Second pass of a K-split layer (or the head sum of the fused attention block): with the consumer's GroupNorm
// finalisation fused when run_ops found one waiting (tl_fin_next) and the shapes allow it.
int run_combine(cm_model *m, cm::CombineArgs &cb, hipStream_t st) {
  static const bool no_fuse = cm::diag_env("CM_NO_FUSE_GNFIN") != nullptr;
  const Op *f = tl_fin_next;
  tl_fin_done = false;
  if (f && !no_fuse && (!f->g1 || f->g1->V() == cb.V)) {
    const Act *g1 = f->g1;
    const int b0 = tl_fin_b0, Ct = cb.C + (g1 ? g1->C : 0);
    cb.fin_gamma = f->gamma; cb.fin_beta = f->beta;
    cb.fin_gn = f->gn_out + (size_t)b0 * 2 * Ct;
    cb.fin_mr = f->gn_mr ? f->gn_mr + (size_t)b0 * 2 * Ct : nullptr;
    cb.fin_p1 = g1 ? g1->part + (size_t)b0 * g1->nslots * g1->C * 2 : nullptr;
    cb.fin_n1 = g1 ? g1->cnt + (size_t)b0 * g1->nslots : nullptr;
    cb.fin_ns1 = g1 ? g1->nslots : 0; cb.fin_C1 = g1 ? g1->C : 0;
    cb.fin_groups = GN_GROUPS; cb.fin_eps = GN_EPS;
    if (cm::combine_gn_ok(cb)) {
      CM_HIP(cm::launch_combine_gn(cb, st));
      tl_fin_done = true;
      return 0;
    }
  }
  CM_HIP(cm::launch_ksplit_combine(cb, st));
  return 0;
}

// Launch the op list for the `B` samples starting at sample `b0` on stream `st`.
// Every sample-indexed pointer is offset by b0, so two disjoint sub-batches can run
// concurrently on two streams (`slab` selects the stream's K-split scratch region).
int run_ops(cm_model *m, int B, hipStream_t st, int b0 = 0, int slab = 0) {
  if (m->astat_all && !m->train_fwd) {
    // accumulator statistics: this forward ADDS to the rows of its samples -- they must be zero.  The sampling loop's step
    // kernel clears them for the next step (no launch); any other caller pays one memset here.
    auto &cl = m->astat_clean[slab & 3];
    if (!(cl.b0 <= b0 && b0 + B <= cl.b0 + cl.B))
      CM_HIP(hipMemsetAsync(m->astat_all + (size_t)b0 * m->astat_C * 3, 0, (size_t)B * m->astat_C * 3 * sizeof(unsigned long long), st));
    cl.B = 0;                                     // dirty from here on
  }
  for (size_t oi = 0; oi < m->ops.size(); ++oi) {
    Op &op = m->ops[oi];
    // the fused attention block runs in the inference plan, its four generic ops in the training forward
    if (op.kind == OP_ATTNBLK ? m->train_fwd : (op.in_attn_block && !m->train_fwd)) continue;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (m->profile) {
      CM_HIP(hipEventCreate(&e0));
      CM_HIP(hipEventCreate(&e1));
      CM_HIP(hipEventRecord(e0, st));
    }
    {
      static const bool sync_ops = cm::diag_env("CM_SYNC_OPS") != nullptr;   // debugging: name every op and drain the device before it
      if (sync_ops) {
        const hipError_t e = hipDeviceSynchronize();
        fprintf(stderr, "[cm] forward reaches op %zu %s (%s)\n", oi, op.label.c_str(), hipGetErrorString(e));
        fflush(stderr);
      }
    }
    // a K-split layer's second pass can carry the GroupNorm finalisation of the op that consumes its output
    tl_fin_next = nullptr; tl_fin_done = false;
    size_t fin_at = 0;
    if ((op.kind == OP_CONV && op.ks > 1) || op.kind == OP_ATTNBLK) {
      const Act *produced = op.kind == OP_CONV ? op.out_act : op.ab_out;
      size_t j = oi + 1;
      while (j < m->ops.size() && (m->ops[j].kind == OP_ATTNBLK ? m->train_fwd : (m->ops[j].in_attn_block && !m->train_fwd))) ++j;
      if (j < m->ops.size() && m->ops[j].kind == OP_GNFIN && m->ops[j].g0 == produced && !(m->ops[j].qr_consumer && !m->train_fwd)) { tl_fin_next = &m->ops[j]; tl_fin_b0 = b0; fin_at = j; }
    }
    switch (op.kind) {
      case OP_CONV:
        if (run_conv(m, op, B, st, b0, slab)) return 1;
        break;
      case OP_STATS: {
        const Act *t = op.act;
        CM_HIP(cm::launch_chan_stats(t->d + (size_t)b0 * t->V() * t->C, B, t->V(), t->C, t->nslice,
                                     t->part + (size_t)b0 * t->nslice * t->C * 2, t->cnt + (size_t)b0 * t->nslice, st));
        const_cast<Act *>(t)->nslots = t->nslice;
        break;
      }
      case OP_GNFIN: {
        if (op.qr_consumer && !m->train_fwd) break;   // its consumer finalises the statistics itself (cm_conv_qr.hip)
        if (op.atomic && !m->train_fwd) break;        // accumulator statistics: finalised by the consuming conv (run_conv)
        op.fin_skipped[slab & 3] = false;
        // (<= 16 slots: measured -0.7 % on the ATC step, -1.6 % on the 24x72 f16 plan; HERMES-CR-120's half resolution has 24 slots
        //  per tensor and up to 192 channels -- there the merge in 256 workgroups cost more than the launch, +0.5 %)
        static const int few_sc = cm::diag_env("CM_FEW_SC") ? atoi(cm::diag_env("CM_FEW_SC")) : 0;   // (slots x channels bound, experiments)
        const auto few = [](const Act *t) { return t->nslots <= 16 || t->nslots * t->C <= few_sc; };
        if (op.from_slots && !m->train_fwd && few(op.g0) && (!op.g1 || few(op.g1)) &&
            (!op.g1 || op.g1->V() == op.g0->V())) {
          op.fin_skipped[slab & 3] = true;            // few slots: the consuming Winograd conv merges them in its prologue (run_conv)
          break;
        }
        { static const bool skip = cm::diag_env("CM_SKIP_GNFIN") != nullptr; if (skip) break; }   // timing bound only (stale rows)
        if (op.g1 && op.g1->V() != op.g0->V()) return fail("concat sources disagree on voxel count");
        const Act *g0 = op.g0, *g1 = op.g1;
        const int Ct = g0->C + (g1 ? g1->C : 0);
        CM_HIP(cm::launch_gn_finalize(g0->part + (size_t)b0 * g0->nslots * g0->C * 2, g0->cnt + (size_t)b0 * g0->nslots,
                                      g0->nslots, g0->C, g1 ? g1->part + (size_t)b0 * g1->nslots * g1->C * 2 : nullptr,
                                      g1 ? g1->cnt + (size_t)b0 * g1->nslots : nullptr, g1 ? g1->nslots : 0,
                                      g1 ? g1->C : 0, g0->V(), op.gamma, op.beta, GN_GROUPS, GN_EPS,
                                      op.gn_out + (size_t)b0 * 2 * Ct, op.gn_mr ? op.gn_mr + (size_t)b0 * 2 * Ct : nullptr, B, st));
        break;
      }
      case OP_ATTN:
        // reduced-precision plan, inference: QK^T and PV on f16 matrix-core operands (the fp32 plan and every training
        // forward keep the exact fp32 kernel)
        if (m->precision == CM_PRECISION_F16 && !m->train_fwd && op.E / ATTN_HEADS == 32 && !cm::diag_env("CM_NO_ATTN_F16"))
          CM_HIP(cm::launch_attn_core_f16(op.qkv + (size_t)b0 * op.S * 3 * op.E, op.aout + (size_t)b0 * op.S * op.E, B, op.S,
                                          op.E, ATTN_HEADS, st));
        else
          CM_HIP(cm::launch_attn_core(op.qkv + (size_t)b0 * op.S * 3 * op.E, op.aout + (size_t)b0 * op.S * op.E, B, op.S,
                                      op.E, ATTN_HEADS, st));
        break;
      case OP_ATTNBLK: {
        const Op &gop = m->ops[op.ab_gn], &qop = m->ops[op.ab_qkv], &oop = m->ops[op.ab_outc];
        float *scratch = m->ks_scratch + (size_t)slab * m->ks_scratch_floats;
        const size_t xoff = (size_t)b0 * op.S * op.E;
        cm::AttnBlockArgs aa{};
        aa.x = op.ab_x->d + xoff; aa.gamma = gop.gamma; aa.beta = gop.beta;
        aa.w_in = op.d_win; aa.b_in = qop.ca.bias; aa.w_out = op.d_wout;
        aa.part = scratch; aa.B = B; aa.S = op.S; aa.E = op.E; aa.heads = ATTN_HEADS; aa.groups = GN_GROUPS; aa.eps = GN_EPS;
        CM_HIP(cm::launch_attn_block(aa, st));
        cm::CombineArgs cb{};
        cb.part = scratch; cb.S = ATTN_HEADS; cb.stride = (long long)B * op.S * op.E;
        cb.bias = oop.ca.bias; cb.temb = nullptr; cb.tidx = m->tbuf;
        cb.resid = aa.x; cb.res_cs = op.E;
        cb.out = op.ab_out->d + xoff; cb.C = op.E; cb.V = op.S; cb.B = B;
        cb.nslots = (op.S + 31) / 32;
        cb.stat_part = op.ab_out->part + (size_t)b0 * cb.nslots * cb.C * 2;
        cb.stat_cnt = op.ab_out->cnt + (size_t)b0 * cb.nslots;
        op.ab_out->nslots = cb.nslots;
        if (run_combine(m, cb, st)) return 1;
        break;
      }
    }
    if (m->profile) {
      CM_HIP(hipEventRecord(e1, st));
      m->prof_events[slab & 3].push_back({(int)oi, {e0, e1}});
    }
    if (m->mid_at >= 0 && (int)oi >= m->mid_at) {
      CM_HIP(hipEventRecord(m->ev_half, st));
      m->mid_at = -1;
    }
    if (tl_fin_done) oi = fin_at;   // (the ops in between are the ones this mode skips anyway)
    tl_fin_next = nullptr; tl_fin_done = false;
  }
  return 0;
}

int prof_begin(cm_model *m, hipStream_t st) {
  if (!m->profile) return 0;
  for (int i = 0; i < K_NCLASS; ++i) { m->prof_ms[i] = 0; m->prof_n[i] = 0; m->prof_union_ms[i] = 0; }
  for (Op &op : m->ops) { op.prof_ms = 0; op.prof_n = 0; }
  if (!m->prof_base) CM_HIP(hipEventCreate(&m->prof_base));
  CM_HIP(hipEventRecord(m->prof_base, st));
  return 0;
}

// Per-launch durations (summed per op and per class) and, per class, the length of the UNION of the launch intervals over
// all batch lanes: with two lanes the launches of one class overlap each other and other classes, so "class time" is the
// time during which at least one launch of the class was running (all offsets against prof_base).
int prof_collect(cm_model *m, hipStream_t st) {
  if (!m->profile) return 0;
  CM_HIP(hipDeviceSynchronize());
  (void)st;
  std::vector<std::pair<float, float>> iv[K_NCLASS];
  for (int ln = 0; ln < 4; ++ln) {
    for (auto &pe : m->prof_events[ln]) {
      float t0 = 0, t1 = 0;
      CM_HIP(hipEventElapsedTime(&t0, m->prof_base, pe.second.first));
      CM_HIP(hipEventElapsedTime(&t1, m->prof_base, pe.second.second));
      const float ms = t1 - t0;
      Op &op = m->ops[pe.first];
      op.prof_ms += ms;
      op.prof_n += 1;
      m->prof_ms[op.cls] += ms;
      m->prof_n[op.cls] += 1;
      iv[op.cls].push_back({t0, t1});
      hipEventDestroy(pe.second.first);
      hipEventDestroy(pe.second.second);
    }
    m->prof_events[ln].clear();
  }
  for (int c = 0; c < K_NCLASS; ++c) {
    std::sort(iv[c].begin(), iv[c].end());
    float total = 0, lo = 0, hi = -1;
    for (auto &x : iv[c]) {
      if (hi < lo || x.first > hi) { if (hi >= lo) total += hi - lo; lo = x.first; hi = x.second; }
      else hi = std::max(hi, x.second);
    }
    if (hi >= lo) total += hi - lo;
    m->prof_union_ms[c] = total;
  }
  return 0;
}

// Makes `dev` current for the duration of one C-ABI call and restores the caller's device on return.
struct DevGuard {
  int prev = -1, cur = -1;
  explicit DevGuard(int dev) : cur(dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) hipSetDevice(dev);
  }
  ~DevGuard() {
    if (prev >= 0 && prev != cur) hipSetDevice(prev);
  }
  DevGuard(const DevGuard &) = delete;
  DevGuard &operator=(const DevGuard &) = delete;
};

int check_ready(cm_model *m, int B) {
  if (!m) return fail("null model handle");
  if (!m->finalized) return fail("cm_model_finalize has not been called");
  if (B < 1 || B > m->cfg.max_batch) return fail("batch %d outside [1, max_batch=%d]", B, m->cfg.max_batch);
  return 0;
}

// torch.linspace / cumprod semantics of forward.py:15-27 (see oracle/unet_numpy.py:
// fp32 step, symmetric fill, fused multiply-add; cumprod accumulated in double).
void build_schedule(cm_schedule *s, int T, float scale, float beta_start, float beta_end) {
  s->T = T;
  for (auto &t : s->tab) t.assign((size_t)T, 0.f);
  const float start = (float)((double)scale * (double)beta_start), end = (float)((double)scale * (double)beta_end);
  const double step = (double)(float)(((double)end - (double)start) / (double)(T - 1));
  double acc = 1.0;
  for (int i = 0; i < T; ++i) {
    const double b = (i < T / 2) ? (double)start + step * i : (double)end - step * (T - 1 - i);
    const float beta = (float)b;
    const float alpha = 1.0f - beta;
    acc *= (double)alpha;
    const float abar = (float)acc;
    s->tab[CM_TAB_BETA][i] = beta;
    s->tab[CM_TAB_ALPHA][i] = alpha;
    s->tab[CM_TAB_ALPHA_BAR][i] = abar;
    s->tab[CM_TAB_SQRT_ALPHA_BAR][i] = sqrtf(abar);
    s->tab[CM_TAB_ONE_BY_SQRT_ALPHA][i] = 1.0f / sqrtf(alpha);
    s->tab[CM_TAB_SQRT_ONE_MINUS_ALPHA_BAR][i] = sqrtf(1.0f - abar);
  }
}

// torch.linspace(start, end, n)[i] in fp32: symmetric fill around the midpoint, one fused multiply-add
// per element (the same restatement as the beta schedule; oracle/unet_numpy.py linspace_f32)
float linspace_f32(float start, float end, int n, int i) {
  if (n == 1) return start;
  const double step = (double)(float)(((double)end - (double)start) / (double)(n - 1));
  const double b = (i < n / 2) ? (double)start + step * i : (double)end - step * (n - 1 - i);
  return (float)b;
}

std::vector<int> visit_order(const cm_schedule *s, const cm_sample_opts *o) {
  std::vector<int> v;
  if (o->sampler == CM_SAMPLER_FM_EULER) {
    // time index per Euler step: (t * TIME_MAX_POS).clamp(0, TIME_MAX_POS - 1).long(), flow_matching.py:214
    const int n = std::max(1, o->fm_steps), tmp = std::max(1, o->fm_time_max_pos);
    for (int i = 0; i < n; ++i) {
      float x = linspace_f32(0.f, 1.f, n, i) * (float)tmp;
      x = std::min(std::max(x, 0.f), (float)(tmp - 1));
      v.push_back((int)x);
    }
  } else if (o->sampler == CM_SAMPLER_DDIM) {
    const int d = std::max(1, o->ddim_divider);
    std::vector<int> taus;
    for (int t = 0; t < s->T - 1; t += d) taus.push_back(t);  // np.arange(0, T-1, divider), ddpm.py:326
    v.assign(taus.rbegin(), taus.rend());
  } else {
    for (int t = s->T - 1; t >= 0; --t) v.push_back(t);       // reversed(range(T)), ddpm.py:214
  }
  if (o->first_steps > 0 && (size_t)o->first_steps < v.size()) v.resize((size_t)o->first_steps);
  return v;
}

}  // namespace

// ================================================================================
// C ABI
// ================================================================================
extern "C" {

const char *cm_last_error(void) { return g_err.c_str(); }
int cm_abi_version(void) { return CM_ABI_VERSION; }

int cm_device_count(int *count) {
  if (!count) return fail("null argument");
  CM_HIP(hipGetDeviceCount(count));
  return 0;
}

int cm_malloc(int device, void **d_ptr, size_t bytes) {
  if (!d_ptr) return fail("null argument");
  DevGuard g(device);
  CM_HIP(hipMalloc(d_ptr, bytes ? bytes : 4));
  return 0;
}
int cm_free(int device, void *d_ptr) {
  DevGuard g(device);
  CM_HIP(hipFree(d_ptr));
  return 0;
}
int cm_memcpy_h2d(int device, void *d_dst, const void *h_src, size_t bytes) {
  DevGuard g(device);
  CM_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
  return 0;
}
int cm_memcpy_d2h(int device, void *h_dst, const void *d_src, size_t bytes) {
  DevGuard g(device);
  CM_HIP(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
  return 0;
}
int cm_memcpy_d2d(int device, void *d_dst, const void *d_src, size_t bytes) {
  DevGuard g(device);
  CM_HIP(hipMemcpy(d_dst, d_src, bytes, hipMemcpyDeviceToDevice));
  return 0;
}
int cm_device_synchronize(int device) {
  DevGuard g(device);
  CM_HIP(hipDeviceSynchronize());
  return 0;
}

int cm_model_create(const cm_unet_config *cfg, cm_model **out) {
  if (!cfg || !out) return fail("null argument");
  if (cfg->n_levels < 1 || cfg->n_levels > CM_MAX_LEVELS) return fail("n_levels %d out of range", cfg->n_levels);
  if (cfg->base_channels < 8 || cfg->base_channels % 8) return fail("base_channels must be a positive multiple of 8");
  if (cfg->in_channels < 1 || cfg->in_channels > 8 || cfg->out_channels < 1 || cfg->out_channels > 8)
    return fail("in/out channels must be in [1,8]");
  if (cfg->max_batch < 1) return fail("max_batch must be >= 1");
  if (cfg->num_res_blocks < 1) return fail("num_res_blocks must be >= 1");
  if ((cfg->base_channels / 2) < 2) return fail("base_channels too small for the sinusoidal table");
  // device < 0: host-only handle -- the state_dict plan (names, shapes, set / get) without a GPU
  // (checkpoint tooling, the sanitizer self-test); cm_model_finalize and everything after it need a device
  if (cfg->device >= 0) {
    int ndev = 0;
    CM_HIP(hipGetDeviceCount(&ndev));
    if (cfg->device >= ndev) return fail("device %d not available (%d devices)", cfg->device, ndev);
  }
  auto m = std::make_unique<cm_model>();
  m->cfg = *cfg;
  m->device = cfg->device;
  for (int l = 0; l < cfg->n_levels; ++l)
    if (cfg->channel_mult[l] < 1 || (cfg->base_channels * cfg->channel_mult[l]) % (GN_GROUPS) != 0)
      return fail("channel_mult[%d] invalid", l);
  build_plan(m.get());
  if (m->device < 0) { *out = m.release(); return 0; }
  DevGuard g(m->device);
  CM_HIP(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  for (int i = 1; i < 4; ++i) {
    CM_HIP(hipStreamCreateWithFlags(&m->lane_stream[i], hipStreamNonBlocking));
    CM_HIP(hipEventCreateWithFlags(&m->ev_join[i], hipEventDisableTiming));
  }
  CM_HIP(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
  CM_HIP(hipEventCreateWithFlags(&m->ev_half, hipEventDisableTiming));
  *out = m.release();
  return 0;
}

int cm_model_destroy(cm_model *m) {
  if (!m) return 0;
  if (m->device < 0) { delete m; return 0; }
  DevGuard g(m->device);
  hipDeviceSynchronize();
  for (void *p : m->allocs) hipFree(p);
  if (m->stream) hipStreamDestroy(m->stream);
  for (int i = 1; i < 4; ++i) {
    if (m->lane_stream[i]) hipStreamDestroy(m->lane_stream[i]);
    if (m->ev_join[i]) hipEventDestroy(m->ev_join[i]);
  }
  if (m->ev_fork) hipEventDestroy(m->ev_fork);
  if (m->ev_half) hipEventDestroy(m->ev_half);
  if (m->prof_base) hipEventDestroy(m->prof_base);
  if (m->train) cm_free_train_state(m->train);
  delete m;
  return 0;
}

int cm_model_num_params(const cm_model *m, int32_t *count) {
  if (!m || !count) return fail("null argument");
  *count = (int32_t)m->params.size();
  return 0;
}

int cm_model_param_info(const cm_model *m, int32_t index, const char **name, int64_t shape[5], int32_t *ndim) {
  if (!m || index < 0 || index >= (int)m->params.size()) return fail("parameter index out of range");
  const Param &p = m->params[index];
  if (name) *name = p.name.c_str();
  if (ndim) *ndim = (int32_t)p.shape.size();
  if (shape)
    for (size_t i = 0; i < 5; ++i) shape[i] = i < p.shape.size() ? p.shape[i] : 1;
  return 0;
}

int cm_model_set_param(cm_model *m, const char *name, const float *h_data, int64_t numel) {
  if (!m || !name || !h_data) return fail("null argument");
  auto it = m->pindex.find(name);
  if (it == m->pindex.end()) return fail("unexpected key in state_dict: %s", name);
  Param &p = m->params[it->second];
  if (numel != p.numel()) return fail("size mismatch for %s: got %lld elements, expected %lld", name, (long long)numel, (long long)p.numel());
  if (m->finalized) return fail("model already finalized; create a new handle to load other weights");
  std::memcpy(p.host.data(), h_data, (size_t)numel * sizeof(float));
  p.set = true;
  return 0;
}

int cm_model_get_param(const cm_model *m, const char *name, float *h_data, int64_t numel) {
  if (!m || !name || !h_data) return fail("null argument");
  auto it = m->pindex.find(name);
  if (it == m->pindex.end()) return fail("unknown parameter %s", name);
  const Param &p = m->params[it->second];
  if (numel != p.numel()) return fail("size mismatch for %s", name);
  std::memcpy(h_data, p.host.data(), (size_t)numel * sizeof(float));
  return 0;
}

int cm_model_set_precision(cm_model *m, int32_t precision) {
  if (!m) return fail("null model handle");
  if (m->finalized) return fail("precision must be chosen before cm_model_finalize");
  if (precision != CM_PRECISION_F32 && precision != CM_PRECISION_F16 && precision != CM_PRECISION_F32R && precision != CM_PRECISION_F32X)
    return fail("unknown precision %d", precision);
  m->precision = precision;
  return 0;
}

int cm_model_finalize(cm_model *m) {
  if (!m) return fail("null model handle");
  if (m->finalized) return 0;
  if (m->device < 0) return fail("host-only handle (device < 0) cannot be finalized: there is no CPU path");
  for (auto &p : m->params)
    if (!p.set) return fail("missing key in state_dict: %s", p.name.c_str());
  DevGuard g(m->device);
  const cm_unet_config &c = m->cfg;
  const size_t B = (size_t)c.max_batch;
  if (dev_alloc(m, (void **)&m->tbuf, B * sizeof(long long))) return 1;
  CM_HIP(hipMemset(m->tbuf, 0, B * sizeof(long long)));
  if (build_ops(m)) return 1;
  if (plan_h16(m)) return 1;
  if (plan_astat(m)) return 1;
  if (plan_slot_consumers(m)) return 1;
  if (build_time_table(m)) return 1;
  const size_t per = (size_t)m->per_sample();
  const size_t per_past = (size_t)c.in_channels * c.rows * c.cols * c.past_len;
  if (dev_alloc(m, (void **)&m->xstate, B * per * sizeof(float))) return 1;
  if (dev_alloc(m, (void **)&m->dropmask, B * (size_t)m->nproj * sizeof(float))) return 1;
  if (dev_alloc(m, (void **)&m->mse_partial, 64 * sizeof(float))) return 1;
  if (dev_alloc(m, (void **)&m->mse_loss, sizeof(float))) return 1;
  if (dev_alloc(m, (void **)&m->stage_fut, B * per * sizeof(float))) return 1;
  if (dev_alloc(m, (void **)&m->stage_out, B * per * sizeof(float))) return 1;
  if (dev_alloc(m, (void **)&m->stage_past, B * per_past * sizeof(float))) return 1;
  CM_HIP(hipDeviceSynchronize());
  m->finalized = true;
  return 0;
}

int cm_unet_forward(cm_model *m, const float *d_future, const int64_t *d_t, const float *d_past, float *d_out,
                    int32_t B, void *stream) {
  if (check_ready(m, B)) return 1;
  if (m->h2_stale && refresh_h2(m)) return 1;
  if (!d_future || !d_t || !d_past || !d_out) return fail("null tensor argument");
  DevGuard g(m->device);
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  const cm_unet_config &c = m->cfg;
  if (prof_begin(m, st)) return 1;
  CM_HIP(hipMemcpyAsync(m->tbuf, d_t, (size_t)B * sizeof(long long), hipMemcpyDeviceToDevice, st));
  CM_HIP(cm::launch_assemble_input(d_past, d_future, m->x8, B, c.in_channels, c.rows, c.cols, c.past_len, c.future_len, 3, st));
  if (run_ops(m, B, st)) return 1;
  CM_HIP(cm::launch_extract_output(m->eps_cl, 8, d_out, B, c.out_channels, c.rows, c.cols, c.past_len, c.future_len, st));
  return prof_collect(m, st);
}

int cm_unet_forward_host(cm_model *m, const float *h_future, const int64_t *h_t, const float *h_past, float *h_out,
                         int32_t B) {
  if (check_ready(m, B)) return 1;
  if (!h_future || !h_t || !h_past || !h_out) return fail("null tensor argument");
  DevGuard g(m->device);
  const cm_unet_config &c = m->cfg;
  for (int i = 0; i < B; ++i)
    if (h_t[i] < 0 || h_t[i] >= TIME_ROWS) return fail("timestep %lld outside [0,%d)", (long long)h_t[i], TIME_ROWS);
  const size_t per = (size_t)m->per_sample();
  const size_t per_past = (size_t)c.in_channels * c.rows * c.cols * c.past_len;
  CM_HIP(hipMemcpy(m->stage_fut, h_future, B * per * sizeof(float), hipMemcpyHostToDevice));
  CM_HIP(hipMemcpy(m->stage_past, h_past, B * per_past * sizeof(float), hipMemcpyHostToDevice));
  long long *dt = nullptr;
  CM_HIP(hipMalloc((void **)&dt, (size_t)B * sizeof(long long)));
  hipError_t e = hipMemcpy(dt, h_t, (size_t)B * sizeof(long long), hipMemcpyHostToDevice);
  int rc = (e != hipSuccess) ? fail("hipMemcpy t failed") : cm_unet_forward(m, m->stage_fut, (const int64_t *)dt, m->stage_past, m->stage_out, B, nullptr);
  if (!rc) {
    e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) rc = fail("forward failed: %s", hipGetErrorString(e));
  }
  hipFree(dt);
  if (rc) return rc;
  CM_HIP(hipMemcpy(h_out, m->stage_out, B * per * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int cm_model_dropout_width(const cm_model *m, int32_t *width) {
  if (!m || !width) return fail("null argument");
  if (!m->finalized) return fail("model not finalized");
  *width = m->nproj;
  return 0;
}

int cm_unet_forward_train(cm_model *m, const float *d_future, const int64_t *d_t, const float *d_past,
                          const float *d_dropmask, float p, uint64_t seed, int64_t sample_id_base, float *d_out,
                          int32_t B, void *stream) {
  if (check_ready(m, B)) return 1;
  if (!d_future || !d_t || !d_past || !d_out) return fail("null tensor argument");
  if (!(p >= 0.f && p < 1.f)) return fail("dropout rate %g outside [0,1)", p);
  DevGuard g(m->device);
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  const cm_unet_config &c = m->cfg;
  if (d_dropmask)
    CM_HIP(hipMemcpyAsync(m->dropmask, d_dropmask, (size_t)B * m->nproj * sizeof(float), hipMemcpyDeviceToDevice, st));
  else
    CM_HIP(cm::launch_dropout_mask(m->dropmask, B, m->nproj, p, seed, sample_id_base, 0, st));
  CM_HIP(hipMemcpyAsync(m->tbuf, d_t, (size_t)B * sizeof(long long), hipMemcpyDeviceToDevice, st));
  CM_HIP(cm::launch_assemble_input(d_past, d_future, m->x8, B, c.in_channels, c.rows, c.cols, c.past_len, c.future_len, 3, st));
  m->train_fwd = true;
  const int rc = run_ops(m, B, st);
  m->train_fwd = false;
  if (rc) return 1;
  CM_HIP(cm::launch_extract_output(m->eps_cl, 8, d_out, B, c.out_channels, c.rows, c.cols, c.past_len, c.future_len, st));
  return 0;
}

int cm_mse_loss(cm_model *m, const float *d_pred, const float *d_target, int64_t n, float *h_loss, void *stream) {
  if (!m || !d_pred || !d_target || !h_loss || n < 1) return fail("bad argument");
  if (!m->finalized) return fail("model not finalized");
  DevGuard g(m->device);
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  CM_HIP(cm::launch_mse_loss(d_pred, d_target, n, m->mse_partial, m->mse_loss, st));
  CM_HIP(hipMemcpyAsync(h_loss, m->mse_loss, sizeof(float), hipMemcpyDeviceToHost, st));
  CM_HIP(hipStreamSynchronize(st));
  return 0;
}

int cm_debug_activation(cm_model *m, const char *name, float *h_out, int64_t capacity, int64_t shape[5]) {
  if (!m || !name || !h_out) return fail("null argument");
  if (!m->finalized) return fail("model not finalized");
  auto it = m->act_by_name.find(name);
  if (it == m->act_by_name.end()) return fail("no activation named %s", name);
  const Act *a = it->second;
  DevGuard g(m->device);
  const int B = m->cfg.max_batch;
  const int C = a->C;
  const int64_t n = (int64_t)B * C * a->V();
  if (capacity < n) return fail("capacity %lld < %lld", (long long)capacity, (long long)n);
  if (a->h16) {
    // f16 tensor of the reduced-precision plan: fetched as stored, converted and re-laid [B,C,H,W,L] on the host (test hook)
    std::vector<uint16_t> raw((size_t)n);
    CM_HIP(hipStreamSynchronize(m->stream));
    CM_HIP(hipMemcpy(raw.data(), a->d, (size_t)n * sizeof(uint16_t), hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b)
      for (int z = 0; z < a->Z; ++z)
        for (int y = 0; y < a->Y; ++y)
          for (int x = 0; x < a->X; ++x)
            for (int c = 0; c < C; ++c)
              h_out[((((size_t)b * C + c) * a->Y + y) * a->X + x) * a->Z + z] =
                  f16_bits_to_f32(raw[((((size_t)b * a->Z + z) * a->Y + y) * a->X + x) * C + c]);
    if (shape) { shape[0] = B; shape[1] = C; shape[2] = a->Y; shape[3] = a->X; shape[4] = a->Z; }
    return 0;
  }
  float *tmp = nullptr;
  CM_HIP(hipMalloc((void **)&tmp, (size_t)n * sizeof(float)));
  hipError_t e = cm::launch_cl_to_ref(a->d, a->C, tmp, B, C, a->Z, a->Y, a->X, m->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
  if (e == hipSuccess) e = hipMemcpy(h_out, tmp, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
  hipFree(tmp);
  if (e != hipSuccess) return fail("debug copy failed: %s", hipGetErrorString(e));
  if (shape) { shape[0] = B; shape[1] = C; shape[2] = a->Y; shape[3] = a->X; shape[4] = a->Z; }
  return 0;
}

// ---- schedule -----------------------------------------------------------------
int cm_schedule_create(int32_t timesteps, float scale, float beta_start, float beta_end, int32_t device, cm_schedule **out) {
  if (!out) return fail("null argument");
  if (timesteps < 2) return fail("timesteps must be >= 2");
  auto s = std::make_unique<cm_schedule>();
  s->device = device;
  build_schedule(s.get(), timesteps, scale, beta_start, beta_end);
  if (device >= 0) {
    DevGuard g(device);
    CM_HIP(hipMalloc((void **)&s->d_sab, (size_t)timesteps * sizeof(float)));
    CM_HIP(hipMalloc((void **)&s->d_s1m, (size_t)timesteps * sizeof(float)));
    CM_HIP(hipMemcpy(s->d_sab, s->tab[CM_TAB_SQRT_ALPHA_BAR].data(), (size_t)timesteps * sizeof(float), hipMemcpyHostToDevice));
    CM_HIP(hipMemcpy(s->d_s1m, s->tab[CM_TAB_SQRT_ONE_MINUS_ALPHA_BAR].data(), (size_t)timesteps * sizeof(float), hipMemcpyHostToDevice));
  }
  *out = s.release();
  return 0;
}

int cm_schedule_destroy(cm_schedule *s) {
  if (!s) return 0;
  if (s->device >= 0) {
    DevGuard g(s->device);
    if (s->d_sab) hipFree(s->d_sab);
    if (s->d_s1m) hipFree(s->d_s1m);
  }
  delete s;
  return 0;
}

int cm_schedule_table(const cm_schedule *s, int32_t which, float *h_out, int32_t capacity) {
  if (!s || !h_out) return fail("null argument");
  if (which < 0 || which > 5) return fail("table id %d out of range", which);
  if (capacity < s->T) return fail("capacity %d < timesteps %d", capacity, s->T);
  std::memcpy(h_out, s->tab[which].data(), (size_t)s->T * sizeof(float));
  return 0;
}

int cm_q_sample(const cm_schedule *s, const float *d_x0, const int64_t *d_t, const float *d_eps, float *d_xt, int32_t B,
                int64_t per_sample, void *stream) {
  if (!s || !d_x0 || !d_t || !d_eps || !d_xt) return fail("null argument");
  if (s->device < 0) return fail("schedule was created host-only (device < 0)");
  DevGuard g(s->device);
  CM_HIP(cm::launch_q_sample(d_x0, (const long long *)d_t, d_eps, s->d_sab, s->d_s1m, d_xt, B, per_sample, (hipStream_t)stream));
  return 0;
}

static void ddpm_coeffs(const cm_schedule *s, int t, float *cx, float *ce, float *cn) {
  const float beta = s->tab[CM_TAB_BETA][t];
  const float c1 = s->tab[CM_TAB_ONE_BY_SQRT_ALPHA][t];
  const float s1m = s->tab[CM_TAB_SQRT_ONE_MINUS_ALPHA_BAR][t];
  *cx = c1;
  *ce = -c1 * (beta / s1m);
  *cn = sqrtf(beta);
}

int cm_ddpm_step(const cm_schedule *s, const float *d_eps, float *d_x, int32_t t, const float *d_noise, uint64_t seed,
                 int64_t sample_id_base, int32_t B, int64_t per_sample, void *stream) {
  if (!s || !d_eps || !d_x) return fail("null argument");
  if (t < 0 || t >= s->T) return fail("timestep %d outside [0,%d)", t, s->T);
  if (s->device < 0) return fail("schedule was created host-only (device < 0)");
  DevGuard g(s->device);
  // eps arrives in reference layout here: express it as a degenerate channels-last
  // tensor with C=1 (cs = 1, P = 0, H = W = 1, F = per_sample).
  cm::StepArgs a{};
  a.x = d_x; a.eps_cl = d_eps; a.cs = 1; a.x8 = nullptr; a.noise = d_noise; a.hist = nullptr;
  a.B = B; a.C = 1; a.H = 1; a.W = 1; a.P = 0; a.F = (int)per_sample;
  ddpm_coeffs(s, t, &a.c_x, &a.c_eps, &a.c_noise);
  a.guid = 0.f; a.seed = seed; a.sample_id_base = sample_id_base; a.step = t; a.draw = t > 0;
  CM_HIP(cm::launch_sampler_step(a, (hipStream_t)stream));
  return 0;
}

int cm_sample_num_steps(const cm_schedule *s, const cm_sample_opts *opts, int32_t *nsteps) {
  if (!s || !opts || !nsteps) return fail("null argument");
  *nsteps = (int32_t)visit_order(s, opts).size();
  return 0;
}

int cm_sample_loop(cm_model *m, const cm_schedule *s, const float *d_past, const float *d_xT, const float *d_noise,
                   const cm_sample_opts *opts, float *d_out, float *d_history, int32_t B, void *stream) {
  if (check_ready(m, B)) return 1;
  if (m->h2_stale && refresh_h2(m)) return 1;
  if (!s || !d_past || !opts || !d_out) return fail("null argument");
  if (s->T > TIME_ROWS) return fail("timesteps %d exceed the %d-row time-embedding table (embeddings.py:7)", s->T, TIME_ROWS);
  if (opts->sampler == CM_SAMPLER_FM_EULER && (opts->fm_steps < 1 || opts->fm_time_max_pos < 1 || opts->fm_time_max_pos > TIME_ROWS))
    return fail("flow-matching sampler needs fm_steps >= 1 and 1 <= fm_time_max_pos <= %d", TIME_ROWS);
  DevGuard g(m->device);
  hipStream_t st = stream ? (hipStream_t)stream : m->stream;
  const cm_unet_config &c = m->cfg;
  const size_t per = (size_t)m->per_sample();
  const std::vector<int> order = visit_order(s, opts);
  if (prof_begin(m, st)) return 1;
  // x_T: injected or drawn on device (ddpm.py:211,242)
  if (d_xT) CM_HIP(hipMemcpyAsync(m->xstate, d_xT, B * per * sizeof(float), hipMemcpyDeviceToDevice, st));
  else CM_HIP(cm::launch_randn(m->xstate, B, (long long)per, opts->seed, opts->sample_id_base, 0x7fffffff, st));
  if (d_history) CM_HIP(hipMemcpyAsync(d_history, m->xstate, B * per * sizeof(float), hipMemcpyDeviceToDevice, st));
  CM_HIP(cm::launch_assemble_input(d_past, m->xstate, m->x8, B, c.in_channels, c.rows, c.cols, c.past_len, c.future_len, 3, st));

  // DDIM carries the schedule values of the previously visited step (ddpm.py:245-248)
  const int last = s->T - 1;
  float beta_t = s->tab[CM_TAB_BETA][last], sab_t = s->tab[CM_TAB_SQRT_ALPHA_BAR][last],
        s1m_t = s->tab[CM_TAB_SQRT_ONE_MINUS_ALPHA_BAR][last];
  // Batch lanes (CM_LANES, default 2): the chains are independent, so the batch is cut in halves that run the whole step
  // sequence on separate streams, each enqueued by its own host thread, with bit-identical results; the ramp-up / tail and the
  // launch gaps of one lane are filled by the other lane's workgroups (-4.5 % per step at B = 64).  A profiled call keeps the
  // lanes: its per-class time is the union of the launch intervals over both lanes (prof_collect).
  static const int want_lanes = getenv("CM_LANES") ? atoi(getenv("CM_LANES")) : 2;
  int lanes = std::max(1, std::min(4, want_lanes));
  if (B < 8 * lanes || stream || opts->use_graph) lanes = 1;
  // a profiled call runs ONE lane unless CM_PROFILE_LANES is set: two host threads recording two events per launch slow the
  // profiled pass itself by ~15 % (measured), which would under-report every kernel class
  static const bool prof_lanes = getenv("CM_PROFILE_LANES") != nullptr;
  if (m->profile && !prof_lanes) lanes = 1;
  int Bl[4], off[4];
  hipStream_t sts[4] = {st, m->lane_stream[1], m->lane_stream[2], m->lane_stream[3]};
  for (int ln = 0, o = 0; ln < lanes; ++ln) {
    Bl[ln] = B / lanes + (ln < B % lanes ? 1 : 0);
    off[ln] = o;
    o += Bl[ln];
  }
  if (lanes > 1) {
    CM_HIP(hipEventRecord(m->ev_fork, st));
    for (int ln = 1; ln < lanes; ++ln) CM_HIP(hipStreamWaitEvent(sts[ln], m->ev_fork, 0));
  }
  // per-step scalars (host): the same numbers drive the eager launches and the graph's device table
  std::vector<cm::StepRow> rows(order.size());
  for (size_t k = 0; k < order.size(); ++k) {
    const int t = order[k];
    cm::StepRow &r = rows[k];
    r.t = t; r.step = t; r.pad = 0;
    if (opts->sampler == CM_SAMPLER_FM_EULER) {
      r.c_x = 1.0f; r.c_eps = (float)(1.0 / (double)opts->fm_steps); r.c_noise = 0.f;  // xt + delta * u, flow_matching.py:219
      r.draw = 0; r.guid = 0.f;
      r.step = (int)k;
    } else if (opts->sampler == CM_SAMPLER_DDIM) {
      const float sab_p = s->tab[CM_TAB_SQRT_ALPHA_BAR][t], s1m_p = s->tab[CM_TAB_SQRT_ONE_MINUS_ALPHA_BAR][t];
      const float sig = opts->ddim_sigma;
      r.c_x = sab_p / sab_t;
      r.c_eps = sqrtf(1.0f - sab_p * sab_p - sig * sig) - sab_p * s1m_t / sab_t;
      r.c_noise = sig;
      r.draw = 1;                                                  // noise on every step (ddpm.py:264)
      r.guid = opts->guidance == CM_GUIDANCE_SPARSITY ? opts->lambda_guidance * sqrtf(beta_t) : 0.f;  // ddpm.py:270
      beta_t = s->tab[CM_TAB_BETA][t]; sab_t = sab_p; s1m_t = s1m_p;
    } else {
      ddpm_coeffs(s, t, &r.c_x, &r.c_eps, &r.c_noise);
      r.draw = t > 0;                                              // ddpm.py:27
      r.guid = opts->guidance == CM_GUIDANCE_SPARSITY ? opts->lambda_guidance * sqrtf(s->tab[CM_TAB_BETA][t]) : 0.f;
    }
  }
  auto base_args = [&]() {
    cm::StepArgs a{};
    a.C = c.in_channels; a.H = c.rows; a.W = c.cols; a.P = c.past_len; a.F = c.future_len;
    a.seed = opts->seed; a.cs = 8;
    return a;
  };
  const bool graph = opts->use_graph && !m->profile && lanes == 1 && !stream && order.size() >= 3;
  if (graph) {
    // hipGraph replay: one captured step (the kernels read their per-step scalars from a device table indexed by a
    // device counter), launched once per remaining step.  Step 0 runs eagerly: it performs the lazy tile set-up
    // and per-kernel attribute calls that may not happen inside a capture.
    if (m->steptab_cap < rows.size()) {
      if (dev_alloc(m, (void **)&m->d_steptab, rows.size() * sizeof(cm::StepRow))) return 1;
      m->steptab_cap = rows.size();
    }
    if (!m->d_kctr && dev_alloc(m, (void **)&m->d_kctr, sizeof(int))) return 1;
    CM_HIP(hipMemcpyAsync(m->d_steptab, rows.data(), rows.size() * sizeof(cm::StepRow), hipMemcpyHostToDevice, st));
    CM_HIP(hipMemsetAsync(m->d_kctr, 0xFF, sizeof(int), st));   // -1
    CM_HIP(hipStreamSynchronize(st));                            // `rows` is host memory of this call
    auto enqueue_step = [&]() -> int {
      CM_HIP(cm::launch_step_begin(m->tbuf, B, m->d_steptab, m->d_kctr, st));
      if (run_ops(m, B, st, 0, 0)) return 1;
      cm::StepArgs al = base_args();
      al.B = B; al.x = m->xstate; al.eps_cl = m->eps_cl; al.x8 = m->x8;
      al.sample_id_base = opts->sample_id_base;
      al.tab = m->d_steptab; al.kctr = m->d_kctr; al.row_stride = (long long)B * per; al.boff = 0;
      al.hist = d_history; al.noise = d_noise;
      if (m->astat_all) { al.zero_u64 = m->astat_all; al.zero_n = (long long)B * m->astat_C * 3; }
      CM_HIP(cm::launch_sampler_step(al, st));
      if (m->astat_all) { m->astat_clean[0].b0 = 0; m->astat_clean[0].B = B; }
      return 0;
    };
    if (enqueue_step()) return 1;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    CM_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    const int rc_cap = enqueue_step();
    hipError_t ec = hipStreamEndCapture(st, &g);
    if (rc_cap || ec != hipSuccess) { if (g) hipGraphDestroy(g); return rc_cap ? 1 : fail("graph capture failed: %s", hipGetErrorString(ec)); }
    ec = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    if (ec != hipSuccess) { hipGraphDestroy(g); return fail("graph instantiate failed: %s", hipGetErrorString(ec)); }
    for (size_t k = 1; k < order.size() && ec == hipSuccess; ++k) ec = hipGraphLaunch(ge, st);
    if (ec == hipSuccess) ec = hipStreamSynchronize(st);
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
    if (ec != hipSuccess) return fail("graph replay failed: %s", hipGetErrorString(ec));
  }
  // steps [k0, k1) of batch lane `ln` on its own stream
  auto lane_steps = [&](int ln, size_t k0, size_t k1) -> int {
    const int b0 = off[ln], Bn = Bl[ln];
    hipStream_t ls = sts[ln];
    for (size_t k = k0; k < k1; ++k) {
      const int t = order[k];
      cm::StepArgs al = base_args();
      const cm::StepRow &r = rows[k];
      al.step = r.step; al.c_x = r.c_x; al.c_eps = r.c_eps; al.c_noise = r.c_noise; al.draw = r.draw; al.guid = r.guid;
      if (k == 0) CM_HIP(cm::launch_fill_t(m->tbuf + b0, Bn, t, ls));
      if (run_ops(m, Bn, ls, b0, ln)) return 1;
      al.B = Bn;
      if (k + 1 < order.size()) { al.t_next = m->tbuf + b0; al.t_next_v = order[k + 1]; }
      al.x = m->xstate + (size_t)b0 * per;
      al.eps_cl = m->eps_cl + (size_t)b0 * m->L() * c.rows * c.cols * 8;
      al.x8 = m->x8 + (size_t)b0 * m->L() * c.rows * c.cols * 8;
      al.sample_id_base = opts->sample_id_base + b0;
      al.hist = d_history ? d_history + (k + 1) * B * per + (size_t)b0 * per : nullptr;
      al.noise = (d_noise && al.draw) ? d_noise + k * B * per + (size_t)b0 * per : nullptr;
      if (m->astat_all) {                          // the step kernel clears this lane's accumulator rows for the next forward
        al.zero_u64 = m->astat_all + (size_t)b0 * m->astat_C * 3;
        al.zero_n = (long long)Bn * m->astat_C * 3;
      }
      CM_HIP(cm::launch_sampler_step(al, ls));
      if (m->astat_all) { m->astat_clean[ln & 3].b0 = b0; m->astat_clean[ln & 3].B = Bn; }
    }
    return 0;
  };
  if (!graph && lanes == 1) {
    if (lane_steps(0, 0, order.size())) return 1;
  } else if (!graph) {
    // Step 0 of every lane from the calling thread (lazy tile set-up and function attributes happen there, once);
    // the remaining steps of lane ln > 0 are enqueued by a host thread of its own, so that the lanes' launch
    // streams fill independently (one thread alternating between the streams is launch-bandwidth bound).
    // Lane offset: lane 1 starts after lane 0 has passed a given op of its first step, so that afterwards one lane's
    // full-resolution (issue-bound) section runs beside the other's quarter-resolution (latency-bound) section instead of both
    // lanes marching through the same section together.  CM_LANE_OFFSET = fraction of the op list (0 = start together).
    static const double lane_off = [] { const char *e = cm::diag_env("CM_LANE_OFFSET"); return e ? atof(e) : 0.0; }();   // (measured: 0.25 ... 0.75 all within noise of 0 -- 1.398 ... 1.414 vs 1.404 / 1.406 ms; off by default)
    for (int ln = 0; ln < lanes; ++ln) {
      if (ln == 0 && lanes == 2 && lane_off > 0.0 && lane_off < 1.0 && order.size() >= 2) m->mid_at = (int)(lane_off * (double)m->ops.size());
      if (ln == 1 && lanes == 2 && lane_off > 0.0 && lane_off < 1.0 && order.size() >= 2) CM_HIP(hipStreamWaitEvent(sts[1], m->ev_half, 0));
      if (lane_steps(ln, 0, 1)) return 1;
      m->mid_at = -1;
    }
    // (nothing may throw across the C ABI: a worker's exception becomes its lane's error; a lane whose thread cannot be
    //  created is enqueued from the calling thread instead)
    std::vector<int> rcs((size_t)lanes, 0);
    std::vector<std::string> errs((size_t)lanes);
    std::vector<std::thread> workers;
    std::vector<char> started((size_t)lanes, 0);
    try {
      workers.reserve((size_t)lanes);
      for (int ln = 1; ln < lanes; ++ln) {
        try {
          workers.emplace_back([&, ln]() {
            try {
              if (hipSetDevice(m->device) != hipSuccess) { rcs[ln] = 1; errs[ln] = "hipSetDevice failed in a lane thread"; return; }
              rcs[ln] = lane_steps(ln, 1, order.size());
              if (rcs[ln]) errs[ln] = g_err;
            } catch (const std::exception &e) {
              rcs[ln] = 1;
              try { errs[ln] = e.what(); } catch (...) {}
            } catch (...) {
              rcs[ln] = 1;
            }
          });
          started[ln] = 1;
        } catch (...) {
          started[ln] = 0;                        // no thread for this lane: the calling thread enqueues it below
        }
      }
      rcs[0] = lane_steps(0, 1, order.size());
      if (rcs[0]) errs[0] = g_err;
      for (int ln = 1; ln < lanes; ++ln)
        if (!started[ln]) {
          rcs[ln] = lane_steps(ln, 1, order.size());
          if (rcs[ln]) errs[ln] = g_err;
        }
    } catch (const std::exception &e) {
      rcs[0] = 1;
      try { errs[0] = e.what(); } catch (...) {}
    } catch (...) {
      rcs[0] = 1;
    }
    for (auto &w : workers) w.join();
    for (int ln = 0; ln < lanes; ++ln)
      if (rcs[ln]) return fail("lane %d: %s", ln, errs[ln].empty() ? "exception in the lane's enqueue thread" : errs[ln].c_str());
  }
  for (int ln = 1; ln < lanes; ++ln) {
    CM_HIP(hipEventRecord(m->ev_join[ln], sts[ln]));
    CM_HIP(hipStreamWaitEvent(st, m->ev_join[ln], 0));
  }
  CM_HIP(hipMemcpyAsync(d_out, m->xstate, B * per * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (opts->check_finite) {
    // sampler-output health check: a NaN / Inf from a bad checkpoint must not come back as rc 0
    if (!m->d_nonfinite && dev_alloc(m, (void **)&m->d_nonfinite, sizeof(int))) return 1;
    CM_HIP(cm::launch_count_nonfinite(m->xstate, (long long)(B * per), m->d_nonfinite, st));
    int bad = 0;
    CM_HIP(hipMemcpyAsync(&bad, m->d_nonfinite, sizeof(int), hipMemcpyDeviceToHost, st));
    CM_HIP(hipStreamSynchronize(st));
    if (prof_collect(m, st)) return 1;
    if (bad) return fail("sampling produced %d non-finite values out of %lld (check_finite)", bad, (long long)(B * per));
    return 0;
  }
  return prof_collect(m, st);
}

int cm_sample_loop_host(cm_model *m, const cm_schedule *s, const float *h_past, const float *h_xT, const float *h_noise,
                        const cm_sample_opts *opts, float *h_out, float *h_history, int32_t B) {
  if (check_ready(m, B)) return 1;
  if (!s || !h_past || !opts || !h_out) return fail("null argument");
  DevGuard g(m->device);
  const cm_unet_config &c = m->cfg;
  const size_t per = (size_t)m->per_sample();
  const size_t per_past = (size_t)c.in_channels * c.rows * c.cols * c.past_len;
  const size_t nsteps = visit_order(s, opts).size();
  CM_HIP(hipMemcpy(m->stage_past, h_past, B * per_past * sizeof(float), hipMemcpyHostToDevice));
  if (h_xT) CM_HIP(hipMemcpy(m->stage_fut, h_xT, B * per * sizeof(float), hipMemcpyHostToDevice));
  float *dn = nullptr, *dh = nullptr;
  if (h_noise) {
    CM_HIP(hipMalloc((void **)&dn, nsteps * B * per * sizeof(float)));
    hipError_t e = hipMemcpy(dn, h_noise, nsteps * B * per * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(dn); return fail("noise upload failed: %s", hipGetErrorString(e)); }
  }
  if (h_history) {
    hipError_t e = hipMalloc((void **)&dh, (nsteps + 1) * B * per * sizeof(float));
    if (e != hipSuccess) { if (dn) hipFree(dn); return fail("history alloc failed: %s", hipGetErrorString(e)); }
  }
  int rc = cm_sample_loop(m, s, m->stage_past, h_xT ? m->stage_fut : nullptr, dn, opts, m->stage_out, dh, B, nullptr);
  if (!rc) {
    hipError_t e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) rc = fail("sampling loop failed: %s", hipGetErrorString(e));
  }
  if (!rc) {
    hipError_t e = hipMemcpy(h_out, m->stage_out, B * per * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess && h_history) e = hipMemcpy(h_history, dh, (nsteps + 1) * B * per * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail("result download failed: %s", hipGetErrorString(e));
  }
  if (dn) hipFree(dn);
  if (dh) hipFree(dh);
  return rc;
}

int cm_profile_enable(cm_model *m, int32_t on) {
  if (!m) return fail("null model handle");
  m->profile = on != 0;
  return 0;
}

int cm_profile_read(cm_model *m, float ms[8], int64_t launches[8]) {
  if (!m || !ms || !launches) return fail("null argument");
  for (int i = 0; i < 8; ++i) { ms[i] = m->prof_ms[i]; launches[i] = m->prof_n[i]; }
  return 0;
}

int cm_profile_read_union(cm_model *m, float ms[8]) {
  if (!m || !ms) return fail("null argument");
  for (int i = 0; i < 8; ++i) ms[i] = m->prof_union_ms[i];
  return 0;
}

int cm_model_cost(const cm_model *m, int32_t B, double *flops, double *bytes) {
  if (!m || !m->finalized) return fail("model not finalized");
  double f = 0, by = 0;
  for (const Op &op : m->ops) {
    if (op.kind == OP_CONV) {
      f += op.flops_per_sample * B;
      const cm::ConvArgs &a = op.ca;
      const double vin = (double)a.Zs * a.Ys * a.Xs, vout = (double)a.Zo * a.Yo * a.Xo;
      by += 4.0 * B * (vin * (a.C0 + a.C1) + vout * a.Co);
    } else if (op.kind == OP_ATTN) {
      f += 4.0 * op.S * (double)op.S * op.E * B;
    }
  }
  double wbytes = 0;
  for (const Param &p : m->params) wbytes += 4.0 * p.numel();
  if (flops) *flops = f;
  if (bytes) *bytes = by + wbytes;
  return 0;
}

int cm_profile_report(cm_model *m, char *buf, int64_t capacity) {
  if (!m || !buf || capacity < 1) return fail("null argument");
  std::string out;
  char line[512];
  for (const Op &op : m->ops) {
    if (op.prof_n == 0) continue;
    const double us = op.prof_ms * 1e3 / op.prof_n;
    if (op.kind == OP_CONV) {
      const cm::ConvArgs &a = op.ca;
      const double tf = op.flops_per_sample * op.prof_B / (us * 1e-6) / 1e12;
      snprintf(line, sizeof(line), "%-52s %9.1f us %7.2f TF %8.1f MF/sample B%d ks%d  %s%d NB%d box %dx%dx%dx%d grid %dx%d CK%d lds %zu\n", op.label.c_str(), us, tf,
               op.flops_per_sample / 1e6, op.prof_B, op.ks,
               "MB", op.MB, op.NB, a.bs, a.bz, a.by, a.bx, a.nts * a.ntz * a.nty * a.ntx,
               (a.Co + 32 * op.NB - 1) / (32 * op.NB), a.CK, cm::conv_lds_bytes(a, op.MB, op.NB));
    } else {
      snprintf(line, sizeof(line), "%-52s %9.1f us\n", op.label.c_str(), us);
    }
    out += line;
  }
  snprintf(buf, (size_t)capacity, "%s", out.c_str());
  return 0;
}

// ---- sampling metrics: the reductions of utils/metrics/metricsGenerator.py on the device ------
int cm_frame_metrics(int32_t device, const float *d_pred, const float *d_gt, int32_t N, int32_t Cc, int32_t H, int32_t W,
                     int32_t F, double *h_out, float *h_minmax) {
  if (!d_pred || !d_gt || !h_out || !h_minmax) return fail("null argument");
  if (N < 1 || Cc < 1 || H < 1 || W < 1 || F < 1) return fail("bad shape");
  DevGuard g(device);
  const size_t cells = (size_t)N * Cc * F;
  double *d_out = nullptr;
  float *d_mm = nullptr;
  CM_HIP(hipMalloc((void **)&d_out, cells * 8 * sizeof(double)));
  hipError_t e = hipMalloc((void **)&d_mm, cells * 2 * sizeof(float));
  if (e == hipSuccess) e = cm::launch_frame_metrics(d_pred, d_gt, N, Cc, H, W, F, d_out, d_mm, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(h_out, d_out, cells * 8 * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(h_minmax, d_mm, cells * 2 * sizeof(float), hipMemcpyDeviceToHost);
  hipFree(d_out);
  if (d_mm) hipFree(d_mm);
  if (e != hipSuccess) return fail("frame metrics failed: %s", hipGetErrorString(e));
  return 0;
}

// ---- tile tuner hooks (tools/tune_tiles.py; not used by the product path) --------------------
int cm_debug_conv_flags(int32_t flags) {
  cm::conv_dbg_override = flags;   // < 0: back to the CM_CONV_DBG environment value
  return 0;
}

int cm_debug_conv_count(const cm_model *m, int32_t *count) {
  if (!m || !m->finalized || !count) return fail("model not finalized");
  *count = (int32_t)m->ops.size();
  return 0;
}

// One line per op: "conv <label> ntaps stride par Ci Co Zo Yo Xo NB MB bz by bx ks flags" or "other <label>".
int cm_debug_conv_info(const cm_model *m, int32_t index, char *buf, int64_t capacity) {
  if (!m || !m->finalized || !buf || index < 0 || index >= (int)m->ops.size()) return fail("bad argument");
  const Op &op = m->ops[index];
  if (op.kind != OP_CONV) { snprintf(buf, (size_t)capacity, "other %s", op.label.c_str()); return 0; }
  const cm::ConvArgs &a = op.ca;
  snprintf(buf, (size_t)capacity, "conv %s %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d", op.label.c_str(), a.ntaps, a.stride, a.par,
           a.C0 + a.C1, a.Co, a.Zo, a.Yo, a.Xo, op.NB, op.MB, a.bz, a.by, a.bx, op.ks,
           (op.small_n ? 1 : 0) | (op.first_k ? 2 : 0) | (op.stat_act ? 4 : 0) | (op.skip_if_fused ? 8 : 0) | (a.CK == 32 ? 16 : 0),
           op.out_act ? op.out_act->C : a.Co);       // (last field: channel stride of the output tensor, cm_debug_conv_io's h_out)
  return 0;
}

// Test hook (tests/test_gpu_six_term_hostile.py): conv op `index` ALONE on the caller's data -- no GroupNorm / SiLU on load, no
// time-embedding row, no residual, no fused skip conv; the bias stays.  h_in0 / h_in1: host, channels-last
// [B][Zs][Ys][Xs][C0 / C1] (h_in1 null when the op has one source); h_out: host [B][Zo][Yo][Xo][channel stride of the output
// tensor].  mode 0: the six-term bf16 form where the plan has one (raw operands are unbounded: never the h2 form); mode 1: the
// same layer on fp32 matrix instructions (the split fragments withheld); mode 2: the h2 form where the plan has one (the caller
// keeps its operands inside the bound the plan guarantees, tests/test_gpu_h2.py).
int cm_debug_conv_io(cm_model *m, int32_t index, int32_t mode, const float *h_in0, const float *h_in1, float *h_out, int32_t B) {
  if (check_ready(m, B)) return 1;
  if (!h_in0 || !h_out || index < 0 || index >= (int)m->ops.size()) return fail("bad argument");
  Op &op = m->ops[index];
  if (op.kind != OP_CONV || !op.in0 || !op.out_act) return fail("op %d is not a convolution", index);
  if (op.tuned_B < 0 && !op.qr) return fail("run a forward first");
  if ((op.in1 != nullptr) != (h_in1 != nullptr)) return fail("op %d has %d source tensors", index, op.in1 ? 2 : 1);
  DevGuard g(m->device);
  hipStream_t st = m->stream;
  const size_t Vs = (size_t)op.in0->V(), Vo = (size_t)op.out_act->V();
  CM_HIP(hipMemcpy(op.in0->d, h_in0, (size_t)B * Vs * op.in0->C * sizeof(float), hipMemcpyHostToDevice));
  if (op.in1) CM_HIP(hipMemcpy(op.in1->d, h_in1, (size_t)B * Vs * op.in1->C * sizeof(float), hipMemcpyHostToDevice));
  Op tmp = op;
  tmp.ca.gn = nullptr; tmp.ca.silu = 0; tmp.ca.temb = nullptr; tmp.temb_off = -1; tmp.ca.resid = nullptr; tmp.resid_act = nullptr;
  tmp.d_s2w = nullptr; tmp.d_wqr_skip = nullptr; tmp.skip_if_fused = false; tmp.dbg_raw = true; tmp.pm_off = -1;
  if (!tmp.qr) tmp.gn_op = -1;
  if (tmp.stat_act && tmp.stat_act->aoff >= 0) tmp.stat_act = nullptr;      // (no additions to the accumulator rows of the plan)
  tmp.dbg_h2 = mode == 2;
  if (mode == 1) { tmp.d_wwino_b6 = nullptr; tmp.d_wqr_b6 = nullptr; tmp.d_wups_b6 = nullptr; tmp.b6d = false; tmp.b6s2 = false; }
  const int ns_keep = op.stat_act ? op.stat_act->nslots : 0;
  const int ns_in_keep = op.in0->nslots;
  if (mode == 2 && op.ups && op.in0->part) {
    // the upsample conv's h2 form takes its per-sample scale from the source tensor's slot statistics: those of the caller's data
    const Act *t = op.in0;
    CM_HIP(cm::launch_chan_stats(t->d, B, t->V(), t->C, t->nslice, t->part, t->cnt, st));
    const_cast<Act *>(t)->nslots = t->nslice;
  }
  const int rc = run_conv(m, tmp, B, st, 0, 0);
  const hipError_t e = hipStreamSynchronize(st);
  if (op.stat_act) op.stat_act->nslots = ns_keep;
  const_cast<Act *>(op.in0)->nslots = ns_in_keep;
  if (rc) return 1;
  if (e != hipSuccess) return fail("debug conv launch failed: %s", hipGetErrorString(e));
  CM_HIP(hipMemcpy(h_out, op.out_act->d, (size_t)B * Vo * op.out_act->C * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

// Average duration (us) of `iters` back-to-back launches of conv op `index` at batch B with the tile
// geometry (MB; bz,by,bx) -- 0 keeps the op's own.  The activations are whatever the last forward left.
int cm_debug_time_conv(cm_model *m, int32_t index, int32_t MB, int32_t bz, int32_t by, int32_t bx, int32_t B,
                       int32_t iters, float *us) {
  if (check_ready(m, B)) return 1;
  if (!us || index < 0 || index >= (int)m->ops.size() || iters < 1) return fail("bad argument");
  Op &op = m->ops[index];
  if (op.kind != OP_CONV || ((op.first_k || op.wino) && MB > 0)) return fail("op %d is not a tunable convolution", index);
  if (op.tuned_B < 0) return fail("run a forward first");
  DevGuard g(m->device);
  hipStream_t st = m->stream;
  const Op saved = op;
  int rc = 0;
  if (MB > 0) {
    cm::ConvArgs &a = op.ca;
    const int osd = a.par ? 2 : 1;
    if (bz < 1 || by < 1 || bx < 1 || (bz * by * bx + 31) / 32 != MB || !cm::conv_variant_exists(MB, op.NB)) rc = fail("invalid geometry");
    if (!rc) {
      a.bs = 1; a.bz = bz; a.by = by; a.bx = bx;
      op.MB = MB;
      a.ntz = (a.Zo / osd + bz - 1) / bz; a.nty = (a.Yo / osd + by - 1) / by; a.ntx = (a.Xo / osd + bx - 1) / bx;
      if (cm::conv_lds_bytes(a, MB, op.NB) > 80 * 1024) rc = fail("tile needs too much LDS");  // two workgroups per CU
      if (!rc && op.small_n && (MB & (MB - 1))) rc = fail("small-N kernel needs a power-of-two MB");
      if (!rc && op.stat_act && a.ntz * a.nty * a.ntx * MB * (a.par ? 8 : 1) > MAX_SLOTS) rc = fail("too many statistics slots");
      if (!rc && cm::conv_halo_voxels(a) > 16384) rc = fail("halo box too large");
    }
    if (!rc) {
      std::vector<int> hv((size_t)cm::conv_halo_voxels(a)), mt((size_t)32 * MB);
      cm::conv_build_tables(a, MB, hv.data(), mt.data());
      hipError_t e = hipMemcpy(op.d_hvtab, hv.data(), hv.size() * sizeof(int), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMemcpy(op.d_mtab, mt.data(), mt.size() * sizeof(int), hipMemcpyHostToDevice);
      if (e != hipSuccess) rc = fail("table upload failed");
    }
  }
  if (!rc) {
    const bool was_skip = op.skip_if_fused;
    op.skip_if_fused = false;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2 && !rc; ++i) rc = run_conv(m, op, B, st, 0, 0);
    hipEventRecord(e0, st);
    for (int i = 0; i < iters && !rc; ++i) rc = run_conv(m, op, B, st, 0, 0);
    hipEventRecord(e1, st);
    hipError_t e = hipStreamSynchronize(st);
    if (!rc && e != hipSuccess) rc = fail("timed launch failed: %s", hipGetErrorString(e));
    float ms = 0.f;
    if (!rc) { hipEventElapsedTime(&ms, e0, e1); *us = ms * 1e3f / (float)iters; }
    hipEventDestroy(e0); hipEventDestroy(e1);
    op.skip_if_fused = was_skip;
  }
  // restore the op's own geometry and tables
  const bool changed = MB > 0;
  op = saved;
  if (changed) {
    std::vector<int> hv((size_t)cm::conv_halo_voxels(op.ca)), mt((size_t)32 * op.MB);
    cm::conv_build_tables(op.ca, op.MB, hv.data(), mt.data());
    CM_HIP(hipMemcpy(op.d_hvtab, hv.data(), hv.size() * sizeof(int), hipMemcpyHostToDevice));
    CM_HIP(hipMemcpy(op.d_mtab, mt.data(), mt.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  return rc;
}

// Matrix-core FLOPs the plan actually EXECUTES per kernel class (<= the algorithmic count: the parity form of the
// upsample convs runs 8 of 27 taps, the Winograd layers 16 multiplies per 2x2 outputs and z tap instead of 36,
// computed on whole 32-row blocks and shifted tiles; padding rows of partly filled tiles are counted as executed).
// `b16`, when given, receives the part of those FLOPs that is issued as 16-bit-operand matrix instructions, in ISSUED FLOPs:
// a six-term layer (fp32 products from exact three-way bf16 splits) issues six v_mfma_f32_32x32x16_bf16 products per
// fp32-equivalent product, an f16-plan layer one; `flops` then keeps the part issued as fp32 matrix instructions.
static int exec_flops_split(const cm_model *m, int32_t B, double flops[8], double *b16) {
  if (!m || !m->finalized || !flops) return fail("model not finalized");
  for (int i = 0; i < 8; ++i) flops[i] = 0;
  if (b16) for (int i = 0; i < 8; ++i) b16[i] = 0;
  const bool p16 = m->precision == CM_PRECISION_F16;
  for (const Op &op : m->ops) {
    if (op.kind == OP_ATTN) { flops[op.cls] += 4.0 * op.S * (double)op.S * op.E * B; continue; }
    if (op.kind != OP_CONV || op.skip_if_fused) continue;
    double mult16 = 0.0;     // 0: fp32 matrix instructions; 1: f16 operands; 6: six-term bf16 products; 3: h2 / relaxed (three cross terms)
    const bool h2l = m->precision == CM_PRECISION_F32 && !m->h2_stale && !op.h2_off;   // (as run_conv)
    const bool rel = m->precision == CM_PRECISION_F32R;
    const cm::ConvArgs &a = op.ca;
    const double Ci = a.C0 + a.C1;
    double f = op.flops_per_sample;
    if (op.f16d && m->precision == CM_PRECISION_F16) {
      const double tiles = (double)(a.Zo / op.f16d_bz) * (a.Yo / op.f16d_by) * (a.Xo / op.f16d_bx);
      f = tiles * 128.0 * op.f16d_mbw * a.Co * (Ci * 27.0 + (op.d_w16d_skip ? op.skip0->C + (op.skip1 ? op.skip1->C : 0) : 0)) * 2;
      mult16 = 1.0;
    } else if (op.fin) {
      // 64-row x 128-column x 32-deep GEMM per (plane, in-plane tile): rows beyond the halo box and columns beyond 27 x Co are padding
      f = (double)(a.Yo / op.fin_by) * (a.Xo / op.fin_bx) * a.Zo * 64.0 * 128.0 * 32.0 * 2;
      mult16 = (p16 && op.d_wfin16) ? 1.0 : ((h2l && op.d_wfin_h2) || rel) ? 3.0 : 6.0;
    } else if ((op.b6d || op.b6s2) && !p16) {
      const double tiles = (double)(a.Zo / op.b6d_bz) * (a.Yo / op.b6d_by) * (a.Xo / op.b6d_bx);
      f = tiles * 32.0 * op.b6d_nw * op.b6d_mbw * a.Co * (Ci * 27.0 + (op.d_wb6d_skip ? op.skip0->C + (op.skip1 ? op.skip1->C : 0) : 0)) * 2;
      mult16 = 6.0;
    } else if (op.qr) {
      {
        // the launcher's own predicate (conv_qr2_b6_ok: 8 groups, channel bound, LDS fit), not a copy of one of its clauses
        cm::QrArgs q{};
        q.C0 = a.C0; q.C1 = a.C1; q.Co = a.Co; q.Y = a.Yo; q.X = a.Xo; q.groups = GN_GROUPS; q.raw = 1; q.wq6 = op.d_wqr_b6;
        if (op.d_wqr_skip) { q.s2w = op.d_wqr_skip; q.s2C0 = op.skip0->C; q.s2C1 = op.skip1 ? op.skip1->C : 0; }
        if (op.d_wqr_b6 && cm::conv_qr2_b6_ok(q)) mult16 = ((h2l && op.d_wqr_h2) || rel) ? 3.0 : 6.0;
      }
      const double rows = 2.0 * (a.Yo * a.Xo > 32 ? 2 : 1) * 32;        // whole 32-row blocks, one or two per plane
      f = rows * a.Co * (Ci * 18.0 + (op.d_wqr_skip ? op.skip0->C + (op.skip1 ? op.skip1->C : 0) : 0)) * 2;
    } else if (op.wino) {
      int bz = 0, by = 0, bx = 0;
      if (cm::conv_wino_pick(a.Zo, a.Yo, a.Xo, &bz, &by, &bx)) {
        const double tiles = (double)(a.Zo / bz) * ((a.Yo + by - 1) / by) * ((a.Xo + bx - 1) / bx);
        f = tiles * ((a.Co + 31) / 32) * 16.0 * 32 * 32 * Ci * 3 * 2;
        if (op.d_s2w) f += tiles * ((a.Co + 31) / 32) * 4.0 * 32 * 32 * (op.skip0->C + (op.skip1 ? op.skip1->C : 0)) * 2;
        if (p16 && op.d_wwino16) mult16 = 1.0;
        else if (!p16 && op.d_wwino_b6 && cm::conv_wino_b6_ok(bz, by, bx, a.Co, a.Zo)) mult16 = ((h2l && op.d_wwino_h2) || rel) ? 3.0 : 6.0;   // (the fused 1x1 skip conv stays fp32: counted with the layer, a few % of it)
      }
    } else if (a.par && op.ups && (op.d_wups16 || !(op.d_wfrag16 && m->precision == CM_PRECISION_F16))) {
      // whole 32-row blocks per (tile, class); planes tiles that span Z skip one of 2 MBW (row block, z tap) pairs
      const double tiles = (double)(a.Zs / op.ups_tz) * (a.Ys / op.ups_ty) * (a.Xs / op.ups_tx);
      const double pairs = 2.0 * op.ups_mbw - ((op.ups_planes && op.ups_tz == a.Zs) ? 1.0 : 0.0);
      f = tiles * 8.0 * 32.0 * pairs * 4.0 * a.Co * Ci * 2;
      if (p16 && op.d_wups16) mult16 = 1.0;
      else if (!p16 && op.d_wups_b6) mult16 = (rel || (h2l && op.d_wups_h2 && !m->astat_all)) ? 3.0 : 6.0;
    } else if (a.par) {
      if (p16 && op.d_wfrag16) mult16 = 1.0;
      f = op.flops_per_sample * 8.0 / 27.0;
      if (cm::conv_zsplit_variant(a, op.MB, op.NB)) f *= 6.0 / 8.0;    // two-plane source: 6 of 8 (row block, z tap) pairs
    } else if (cm::conv_zsplit_variant(a, op.MB, op.NB)) {
      f *= 18.0 / 27.0;                                                // two-plane grid: the padding-plane tap is never issued
    }
    if (b16 && mult16 > 0.0) b16[op.cls] += mult16 * f * B;
    else flops[op.cls] += f * B;
  }
  return 0;
}

int cm_model_exec_flops(const cm_model *m, int32_t B, double flops[8]) { return exec_flops_split(m, B, flops, nullptr); }

int cm_model_issue_flops(const cm_model *m, int32_t B, double f32[8], double b16[8]) {
  if (!b16) return fail("null output");
  return exec_flops_split(m, B, f32, b16);
}

int cm_model_class_flops(const cm_model *m, int32_t B, double flops[8]) {
  if (!m || !m->finalized || !flops) return fail("model not finalized");
  for (int i = 0; i < 8; ++i) flops[i] = 0;
  for (const Op &op : m->ops) {
    if (op.kind == OP_CONV) flops[op.cls] += op.flops_per_sample * B;
    else if (op.kind == OP_ATTN) flops[op.cls] += 4.0 * op.S * (double)op.S * op.E * B;
  }
  return 0;
}

}  // extern "C"

#include "cm_train_host.inc"
