// Host-side self-test of libcrowdmod_hip's C++ (`make asan`): the plan builder, the weight / index
// packers, the tile planner with its coordinate tables, the schedule and the error paths, compiled
// under AddressSanitizer + UBSan and run WITHOUT a GPU ("sanitizers": GPU ASan is
// not available on this pool, so the pointer-heavy host code is what gets sanitised).  It includes
// cm_model.cpp itself to reach the internal functions; no kernel is ever launched.
#include "cm_model.cpp"

#include <cstdlib>
#include <random>

namespace {

int g_checks = 0;
#define EXPECT(cond)                                                                  \
  do {                                                                                \
    ++g_checks;                                                                       \
    if (!(cond)) { fprintf(stderr, "selftest FAILED: %s (%s:%d)\n", #cond, __FILE__, __LINE__); exit(1); } \
  } while (0)

void test_schedule() {
  for (int T : {2, 3, 50, 1000}) {
    cm_schedule *s = nullptr;
    EXPECT(cm_schedule_create(T, 0.5f, 1e-4f, 2e-2f, -1, &s) == 0);
    std::vector<float> buf((size_t)T);
    for (int w = 0; w < 6; ++w) EXPECT(cm_schedule_table(s, w, buf.data(), T) == 0);
    EXPECT(cm_schedule_table(s, 6, buf.data(), T) != 0);
    EXPECT(cm_schedule_table(s, 0, buf.data(), T - 1) != 0);
    cm_sample_opts o{};
    int32_t n = 0;
    o.sampler = CM_SAMPLER_DDPM;
    EXPECT(cm_sample_num_steps(s, &o, &n) == 0 && n == T);
    for (int d : {1, 2, 7, 100, 5000}) {
      o.sampler = CM_SAMPLER_DDIM; o.ddim_divider = d;
      EXPECT(cm_sample_num_steps(s, &o, &n) == 0 && n == (T - 1 + d - 1) / d);
    }
    o.sampler = CM_SAMPLER_FM_EULER; o.fm_steps = 10; o.fm_time_max_pos = 1000;
    EXPECT(cm_sample_num_steps(s, &o, &n) == 0 && n == 10);
    o.first_steps = 3;
    EXPECT(cm_sample_num_steps(s, &o, &n) == 0 && n == 3);
    // a host-only schedule must refuse device work instead of dereferencing null tables
    EXPECT(cm_q_sample(s, buf.data(), (const int64_t *)buf.data(), buf.data(), buf.data(), 1, 1, nullptr) != 0);
    EXPECT(cm_schedule_destroy(s) == 0);
  }
  cm_schedule *s = nullptr;
  EXPECT(cm_schedule_create(1, 0.5f, 1e-4f, 2e-2f, -1, &s) != 0);
  EXPECT(cm_schedule_create(10, 0.5f, 1e-4f, 2e-2f, -1, nullptr) != 0);
}

cm_unet_config atc_cfg(int C, int rows, int cols, int device) {
  cm_unet_config c{};
  c.in_channels = c.out_channels = C;
  c.num_res_blocks = 1; c.base_channels = 32; c.n_levels = 3;
  const int mult[3] = {1, 2, 4}, att[3] = {0, 0, 1};
  for (int i = 0; i < 3; ++i) { c.channel_mult[i] = mult[i]; c.apply_attention[i] = att[i]; }
  c.time_multiple = 4; c.rows = rows; c.cols = cols; c.past_len = 5; c.future_len = 3; c.max_batch = 2; c.device = device;
  return c;
}

void test_plan_and_params() {
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  for (int C : {3, 4}) {
    cm_unet_config c = atc_cfg(C, 12, 36, -1);
    cm_model *m = nullptr;
    EXPECT(cm_model_create(&c, &m) == 0);
    int32_t n = 0;
    EXPECT(cm_model_num_params(m, &n) == 0 && n == 169);
    for (int i = 0; i < n; ++i) {
      const char *name = nullptr; int64_t shape[5]; int32_t nd = 0;
      EXPECT(cm_model_param_info(m, i, &name, shape, &nd) == 0 && name && nd >= 1 && nd <= 5);
      int64_t numel = 1;
      for (int d = 0; d < nd; ++d) numel *= shape[d];
      std::vector<float> w((size_t)numel), back((size_t)numel);
      for (auto &v : w) v = U(rng);
      EXPECT(cm_model_set_param(m, name, w.data(), numel) == 0);
      EXPECT(cm_model_set_param(m, name, w.data(), numel + 1) != 0);
      EXPECT(cm_model_get_param(m, name, back.data(), numel) == 0 && back == w);
    }
    EXPECT(cm_model_param_info(m, n, nullptr, nullptr, nullptr) != 0);
    EXPECT(cm_model_set_param(m, "no.such.tensor", (const float *)&n, 1) != 0);
    EXPECT(cm_model_finalize(m) != 0);                     // host-only: no CPU path
    EXPECT(std::string(cm_last_error()).find("host-only") != std::string::npos);
    EXPECT(cm_unet_forward(m, nullptr, nullptr, nullptr, nullptr, 1, nullptr) != 0);
    EXPECT(cm_model_destroy(m) == 0);
  }
  cm_unet_config bad = atc_cfg(3, 12, 36, -1);
  cm_model *m = nullptr;
  bad.base_channels = 12; EXPECT(cm_model_create(&bad, &m) != 0);
  bad = atc_cfg(3, 12, 36, -1); bad.n_levels = 9; EXPECT(cm_model_create(&bad, &m) != 0);
  bad = atc_cfg(9, 12, 36, -1); EXPECT(cm_model_create(&bad, &m) != 0);
  bad = atc_cfg(3, 12, 36, -1); bad.max_batch = 0; EXPECT(cm_model_create(&bad, &m) != 0);
  EXPECT(cm_model_create(nullptr, &m) != 0);
}

void test_packers() {
  std::mt19937 rng(2);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  struct Case { int Co, Ci, ntaps, Ci_pad, CK, NB; };
  const Case cases[] = {{32, 32, 27, 32, 32, 1}, {64, 96, 27, 96, 32, 2}, {32, 3, 27, 8, 8, 1}, {4, 32, 27, 32, 32, 1},
                        {384, 128, 1, 128, 128, 2}, {128, 192, 1, 192, 64, 2}, {40, 24, 27, 24, 8, 2}};
  for (const Case &c : cases) {
    std::vector<float> W((size_t)c.Co * c.Ci * c.ntaps);
    for (auto &v : W) v = U(rng);
    const std::vector<float> wi = to_internal_taps(W.data(), c.Co, c.Ci, c.ntaps);
    EXPECT(wi.size() == W.size());
    const std::vector<float> wf = pack_conv_weights(wi.data(), c.Co, c.Ci, c.ntaps, c.Ci_pad, c.CK, c.NB);
    const int TN = 32 * c.NB, ntn = (c.Co + TN - 1) / TN;
    EXPECT(wf.size() == (size_t)ntn * (c.Ci_pad / c.CK) * c.ntaps * (c.CK / 8) * c.NB * 256);
    // every reference weight appears exactly once; everything else is zero padding
    double sa = 0, sb = 0;
    for (float v : wi) sa += std::fabs(v);
    for (float v : wf) sb += std::fabs(v);
    EXPECT(std::fabs(sa - sb) <= 1e-6 * sa + 1e-9);
    if (c.ntaps == 27) {
      const std::vector<float> wp = parity_weights(wi, c.Co, c.Ci);
      EXPECT(wp.size() == (size_t)8 * c.Co * c.Ci * 8);
      // each parity class redistributes all 27 taps: the tap sum per (co, ci) is preserved
      for (int p = 0; p < 8; ++p) {
        double s27 = 0, s8 = 0;
        for (int t = 0; t < 27; ++t) s27 += wi[t];
        for (int e = 0; e < 8; ++e) s8 += wp[(size_t)p * c.Co * c.Ci * 8 + e];
        EXPECT(std::fabs(s27 - s8) < 1e-5);
      }
    }
    // index version (training re-pack): same positions are filled
    std::vector<int> src((size_t)c.Co * c.Ci * c.ntaps);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (int)i;
    const std::vector<int> pi = pack_conv_indices(src, 1, c.Co, c.Ci, c.ntaps, c.Ci_pad, c.CK, c.NB);
    EXPECT(pi.size() == wf.size());
    for (size_t i = 0; i < pi.size(); ++i) {
      EXPECT(pi[i] >= -1 && pi[i] < (int)src.size());
      if (pi[i] >= 0) EXPECT(wf[i] == wi[(size_t)pi[i]]); else EXPECT(wf[i] == 0.f);
    }
  }
}

void check_tables(const cm::ConvArgs &a, int MB) {
  const int HZ = (a.bz - 1) * a.stride + a.td, HY = (a.by - 1) * a.stride + a.td, HX = (a.bx - 1) * a.stride + a.td;
  const int hv = cm::conv_halo_voxels(a);
  EXPECT(hv == a.bs * HZ * HY * HX);
  std::vector<int> hvt((size_t)hv), mt((size_t)32 * MB);
  cm::conv_build_tables(a, MB, hvt.data(), mt.data());
  for (int v : hvt) {
    EXPECT((v & 511) < HX && ((v >> 9) & 511) < HY && ((v >> 18) & 255) < HZ && (v >> 26) < a.bs);
  }
  int valid = 0;
  for (int v : mt) {
    if (v < 0) continue;
    ++valid;
    EXPECT((v & 511) < a.bx && ((v >> 9) & 511) < a.by && ((v >> 18) & 255) < a.bz && (v >> 26) < a.bs);
  }
  EXPECT(valid == a.bs * a.bz * a.by * a.bx && valid <= 32 * MB);
  EXPECT(cm::conv_lds_bytes(a, MB, 1) > 0);
}

void test_tile_planner() {
  // every 3x3x3 / 1x1x1 layer shape of the three reference grids, through the same chooser the model uses
  struct Grid { int Z, Y, X; };
  const Grid grids[] = {{8, 12, 36}, {8, 28, 24}, {8, 24, 72}, {8, 4, 8}};
  for (const Grid &g : grids)
    for (int level = 0; level < 3; ++level)
      for (int ci : {32, 64, 128, 192, 256})
        for (int co : {32, 64, 128})
          for (int mode = 0; mode < 4; ++mode) {   // 0: 3x3x3 s1, 1: stride 2, 2: parity upsample, 3: 1x1x1
            Op op;
            op.kind = OP_CONV;
            cm::ConvArgs &a = op.ca;
            const int Z = g.Z >> level, Y = g.Y >> level, X = g.X >> level;
            if (Z < 1 || Y < 1 || X < 1) continue;
            if (mode == 1 && (Z < 2 || Y < 2 || X < 2)) continue;
            a.C0 = ci; a.C1 = 0; a.Co = co; a.CK = mode == 3 ? 64 : 32;
            if (ci % a.CK) continue;
            a.ntaps = mode == 3 ? 1 : (mode == 2 ? 8 : 27);
            a.td = mode == 3 ? 1 : (mode == 2 ? 2 : 3);
            a.stride = mode == 1 ? 2 : 1; a.par = mode == 2; a.ups = 0;
            a.Zs = Z; a.Ys = Y; a.Xs = X;
            const int os = mode == 2 ? 2 : 1;
            a.Zo = mode == 1 ? Z / 2 : Z * os; a.Yo = mode == 1 ? Y / 2 : Y * os; a.Xo = mode == 1 ? X / 2 : X * os;
            op.NB = co > 32 ? 2 : 1;
            static Act dummy;
            op.stat_act = &dummy;
            pick_tile(op, TUNE_BATCH);
            EXPECT(op.MB >= 1 && op.MB <= 8 && cm::conv_variant_exists(op.MB, op.NB));
            EXPECT(a.bs >= 1 && a.bz >= 1 && a.by >= 1 && a.bx >= 1);
            EXPECT(a.bs * a.bz * a.by * a.bx <= 32 * op.MB);
            EXPECT(a.ntz * a.bz >= a.Zo / os && a.nty * a.by >= a.Yo / os && a.ntx * a.bx >= a.Xo / os);
            EXPECT(cm::conv_lds_bytes(a, op.MB, op.NB) <= 160 * 1024);
            check_tables(a, op.MB);
          }
}

// round 3: packers and planners of the new kernels (whole-sample quarter-resolution kernel, direct f16 kernel)
void test_round3_packers() {
  std::mt19937 rng(5);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  for (int Co : {32, 64, 128})
    for (int Ci : {16, 64, 192}) {
      std::vector<float> wi((size_t)Co * Ci * 27);
      for (auto &v : wi) v = U(rng);
      // pack_qr: every weight appears exactly once, at the documented position
      const std::vector<float> q = pack_qr(wi, Co, Ci);
      EXPECT(q.size() == wi.size());
      const int ng = 9 * (Ci / 8);
      for (int probe = 0; probe < 200; ++probe) {
        const int co = (int)(rng() % Co), ci = (int)(rng() % Ci), dz = (int)(rng() % 3), dy = (int)(rng() % 3), dx = (int)(rng() % 3);
        const int nt = co / 32, g = (ci / 8) * 9 + dy * 3 + dx, lane = 32 * ((ci % 8) / 4) + co % 32, jj = ci % 4;
        EXPECT(q[((((size_t)nt * ng + g) * 3 + dz) * 64 + lane) * 4 + jj] == wi[((size_t)co * Ci + ci) * 27 + (dz * 3 + dy) * 3 + dx]);
      }
      std::vector<float> w2((size_t)Co * Ci);
      for (auto &v : w2) v = U(rng);
      const std::vector<float> qs = pack_qr_skip(w2.data(), Co, Ci);
      EXPECT(qs.size() == w2.size());
      for (int probe = 0; probe < 100; ++probe) {
        const int co = (int)(rng() % Co), ci = (int)(rng() % Ci);
        EXPECT(qs[(((size_t)(co / 32) * (Ci / 8) + ci / 8) * 64 + 32 * ((ci % 8) / 4) + co % 32) * 4 + ci % 4] == w2[(size_t)co * Ci + ci]);
      }
      // pack_f16d: two halves per float, [nt][c16][tap][nb][lane][8]
      for (int NB : {1, 2}) {
        if (Co % (32 * NB) || Ci % 16) continue;
        const std::vector<float> f = pack_f16d(wi.data(), Co, Ci, 27, NB);
        EXPECT(f.size() * 2 == wi.size());
        const uint16_t *h = reinterpret_cast<const uint16_t *>(f.data());
        for (int probe = 0; probe < 100; ++probe) {
          const int co = (int)(rng() % Co), ci = (int)(rng() % Ci), t = (int)(rng() % 27);
          const int nt = co / (32 * NB), nb = (co % (32 * NB)) / 32, c = ci / 16, lane = 32 * ((ci % 16) / 8) + co % 32, j = ci % 8;
          const size_t o = (((((size_t)nt * (Ci / 16) + c) * 27 + t) * NB + nb) * 64 + lane) * 8 + j;
          EXPECT(h[o] == f32_to_f16_bits(wi[((size_t)co * Ci + ci) * 27 + t]));
        }
      }
      // pack_ups_b6: hi + mid + lo reproduces every weight EXACTLY (8 + 8 + 8 mantissa bits), at the documented positions
      if (Ci % 32 == 0) {
        std::vector<float> w8((size_t)Co * Ci * 8);
        for (auto &v : w8) v = U(rng) * (rng() % 7 == 0 ? 1e-6f : 1.f);
        const std::vector<float> f = pack_ups_b6(w8.data(), Co, Ci);
        EXPECT(f.size() * 2 == w8.size() * 3);
        const uint16_t *h = reinterpret_cast<const uint16_t *>(f.data());
        for (int probe = 0; probe < 200; ++probe) {
          const int co = (int)(rng() % Co), ci = (int)(rng() % Ci), t = (int)(rng() % 8);
          const int cbk = co / 32, ch = ci / 32, mg = (ci % 32) / 16, lane = 32 * ((ci % 16) / 8) + co % 32, i = ci % 8;
          float sum = 0.f;
          for (int tm = 2; tm >= 0; --tm)
            sum += bf16_bits_to_f32(h[(((((((size_t)cbk * (Ci / 32) + ch) * 8 + t) * 2 + mg) * 3 + tm) * 64) + lane) * 8 + i]);
          EXPECT(sum == w8[((size_t)co * Ci + ci) * 8 + t]);
        }
      }
      // pack_ups_f16 (one parity class, [Co][Ci][8]): [column block][chunk][tap][16-channel group][lane][8]
      if (Ci % 32 == 0) {
        std::vector<float> w8((size_t)Co * Ci * 8);
        for (auto &v : w8) v = U(rng);
        const std::vector<float> f = pack_ups_f16(w8.data(), Co, Ci);
        EXPECT(f.size() * 2 == w8.size());
        const uint16_t *h = reinterpret_cast<const uint16_t *>(f.data());
        for (int probe = 0; probe < 100; ++probe) {
          const int co = (int)(rng() % Co), ci = (int)(rng() % Ci), t = (int)(rng() % 8);
          const int cbk = co / 32, ch = ci / 32, mg = (ci % 32) / 16, lane = 32 * ((ci % 16) / 8) + co % 32, i = ci % 8;
          const size_t o = ((((((size_t)cbk * (Ci / 32) + ch) * 8 + t) * 2 + mg) * 64) + lane) * 8 + i;
          EXPECT(h[o] == f32_to_f16_bits(w8[((size_t)co * Ci + ci) * 8 + t]));
        }
      }
    }
  // tile planner of the direct f16 kernel: every reference grid level gets a tile that divides it and fits the kernel's limits
  const int grids[][3] = {{8, 12, 36}, {4, 6, 18}, {8, 28, 24}, {4, 14, 12}, {8, 24, 72}, {4, 12, 36}};
  for (auto &g : grids) {
    int bz = 0, by = 0, bx = 0, mbw = 0;
    EXPECT(cm::conv_f16d_pick(g[0], g[1], g[2], &bz, &by, &bx, &mbw));
    EXPECT(g[0] % bz == 0 && g[1] % by == 0 && g[2] % bx == 0 && bz * by * bx <= 128 * mbw && mbw >= 1 && mbw <= 2);
    cm::ConvArgs a{};
    a.ntaps = 27; a.stride = 1; a.C0 = 32; a.Co = 32; a.Zs = a.Zo = g[0]; a.Ys = a.Yo = g[1]; a.Xs = a.Xo = g[2]; a.bz = bz; a.by = by; a.bx = bx;
    EXPECT(cm::conv_f16d_ok(a, mbw) && cm::conv_f16d_slots(a, mbw) <= MAX_SLOTS);
  }
  // direct six-term kernel (cm_conv_b6d.hip): tile picker, staging lists and row tables on the full-resolution grids of the
  // reference configs -- every real halo voxel staged exactly once into its own LDS row, every output voxel owned by exactly one
  // row, tap reads inside the image, and a row order whose ds_read_b128 service groups are conflict-free
  {
    const int fgrids[][3] = {{8, 12, 36}, {8, 28, 24}, {8, 24, 72}, {16, 12, 36}, {8, 8, 12}, {8, 12, 20}, {4, 6, 18}};
    for (auto &g : fgrids) {
      int bz = 0, by = 0, bx = 0, nw = 0, mbw = 0;
      EXPECT(cm::conv_b6d_pick(g[0], g[1], g[2], &bz, &by, &bx, &nw, &mbw));
      cm::ConvArgs a{};
      a.ntaps = 27; a.stride = 1; a.C0 = 32; a.Co = 32; a.Zs = a.Zo = g[0]; a.Ys = a.Yo = g[1]; a.Xs = a.Xo = g[2]; a.bz = bz; a.by = by; a.bx = bx;
      EXPECT(cm::conv_b6d_ok(a, nw, mbw) && cm::conv_b6d_slots(a, nw, mbw) <= MAX_SLOTS);
      std::vector<int> tS, tM;
      int NSP = 0, HVP = 0, PY = 0, PZ = 0, ntp = 0, conf = -1;
      cm::conv_b6d_tables(g[0], g[1], g[2], bz, by, bx, nw, mbw, tS, tM, &NSP, &HVP, &PY, &PZ, &ntp, &conf);
      const int HV = (bz + 2) * (by + 2) * (bx + 2), MR = 32 * nw * mbw, V = g[0] * g[1] * g[2];
      EXPECT(ntp == (g[0] / bz) * (g[1] / by) * (g[2] / bx) && HVP >= HV && (HVP & 7) == 4 && 4 * NSP <= 10 * 64 * nw);
      if (g[0] == 8 && g[1] % 4 == 0 && g[2] % 4 == 0) EXPECT(conf == 0);
      std::vector<int> owned((size_t)V, 0);
      for (int p = 0; p < ntp; ++p) {
        std::vector<char> rowseen((size_t)HV, 0);
        for (int i = 0; i < NSP; ++i) {
          const int src = tS[((size_t)p * NSP + i) * 2], row = tS[((size_t)p * NSP + i) * 2 + 1];
          if (src < 0) continue;
          EXPECT(src < V && row >= 0 && row < HV && !rowseen[(size_t)row]);
          rowseen[(size_t)row] = 1;
        }
        for (int m2 = 0; m2 < MR; ++m2) {
          const int hidx = tM[((size_t)p * MR + m2) * 2], ov = tM[((size_t)p * MR + m2) * 2 + 1];
          EXPECT(hidx >= 0 && hidx + 2 * PZ + 2 * PY + 2 < HVP);
          if (ov >= 0) { EXPECT(ov < V); owned[(size_t)ov] += 1; EXPECT(rowseen[(size_t)(hidx + PZ + PY + 1)]); }   // its centre tap is a staged voxel
        }
      }
      for (int v = 0; v < V; ++v) EXPECT(owned[(size_t)v] == 1);
    }
  }
  // the same kernel at stride 2 (the DownSample convs of the reference configs, source grid -> output grid): every tap of every
  // output voxel reads the LDS row that holds exactly source voxel 2 o - 1 + d, or a never-written (zero) row when that is padding
  {
    const int sgrids[][3] = {{8, 12, 36}, {4, 6, 18}, {8, 28, 24}, {4, 14, 12}, {8, 24, 72}, {4, 12, 36}, {16, 12, 36}, {8, 12, 20}, {4, 6, 10}, {5, 7, 9}};
    for (auto &g : sgrids) {
      const int Zo = (g[0] - 1) / 2 + 1, Yo = (g[1] - 1) / 2 + 1, Xo = (g[2] - 1) / 2 + 1;
      int bz = 0, by = 0, bx = 0, nw = 0, mbw = 0;
      EXPECT(cm::conv_b6d_pick(Zo, Yo, Xo, &bz, &by, &bx, &nw, &mbw, 2, g[0], g[1], g[2]));
      cm::ConvArgs a{};
      a.ntaps = 27; a.stride = 2; a.C0 = 32; a.Co = 32; a.Zs = g[0]; a.Ys = g[1]; a.Xs = g[2]; a.Zo = Zo; a.Yo = Yo; a.Xo = Xo; a.bz = bz; a.by = by; a.bx = bx;
      EXPECT(cm::conv_b6d_ok(a, nw, mbw) && cm::conv_b6d_slots(a, nw, mbw) <= MAX_SLOTS && mbw == 1);
      std::vector<int> tS, tM;
      int NSP = 0, HVP = 0, PY = 0, PZ = 0, ntp = 0, conf = -1;
      cm::conv_b6d_tables(Zo, Yo, Xo, bz, by, bx, nw, mbw, tS, tM, &NSP, &HVP, &PY, &PZ, &ntp, &conf, 2, g[0], g[1], g[2]);
      const int HV = (2 * bz + 1) * (2 * by + 1) * (2 * bx + 1), MR = 32 * nw * mbw, Vo = Zo * Yo * Xo, Vs = g[0] * g[1] * g[2];
      EXPECT(ntp == (Zo / bz) * (Yo / by) * (Xo / bx) && HVP >= HV && (HVP & 7) == 4 && 4 * NSP <= cm::conv_b6d_nld(2) * 64 * nw);
      std::vector<int> owned((size_t)Vo, 0);
      for (int p = 0; p < ntp; ++p) {
        std::vector<int> rowsrc((size_t)HV, -1);
        for (int i = 0; i < NSP; ++i) {
          const int src = tS[((size_t)p * NSP + i) * 2], row = tS[((size_t)p * NSP + i) * 2 + 1];
          if (src < 0) continue;
          EXPECT(src < Vs && row >= 0 && row < HV && rowsrc[(size_t)row] < 0);
          rowsrc[(size_t)row] = src;
        }
        for (int m2 = 0; m2 < MR; ++m2) {
          const int hidx = tM[((size_t)p * MR + m2) * 2], ov = tM[((size_t)p * MR + m2) * 2 + 1];
          EXPECT(hidx >= 0 && hidx + 2 * PZ + 2 * PY + 2 < HVP);
          if (ov < 0) continue;
          EXPECT(ov < Vo);
          owned[(size_t)ov] += 1;
          const int oz = ov / (Yo * Xo), oy = (ov / Xo) % Yo, ox = ov % Xo;
          for (int t = 0; t < 27; ++t) {
            const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
            const int cz = 2 * oz - 1 + dz, cy = 2 * oy - 1 + dy, cx = 2 * ox - 1 + dx;
            const bool real = cz >= 0 && cz < g[0] && cy >= 0 && cy < g[1] && cx >= 0 && cx < g[2];
            const int row = hidx + dz * PZ + dy * PY + dx;
            EXPECT(row < HV && rowsrc[(size_t)row] == (real ? (cz * g[1] + cy) * g[2] + cx : -1));
          }
        }
      }
      for (int v = 0; v < Vo; ++v) EXPECT(owned[(size_t)v] == 1);
    }
  }
  // stage-once upsample kernel: a source tile for every upsample source grid of the reference configs (ATC, CR-120, 24x72),
  // planes tiles where a plane fits one row block; the launcher's own feasibility test agrees with the picker
  const int srcs[][3] = {{2, 3, 9}, {4, 6, 18}, {2, 7, 6}, {4, 14, 12}, {2, 6, 18}, {4, 12, 36}};
  for (auto &g : srcs) {
    int tz = 0, ty = 0, tx = 0, mbw = 0, planes = 0;
    EXPECT(cm::conv_ups_pick(g[0], g[1], g[2], &tz, &ty, &tx, &mbw, &planes));
    EXPECT(g[0] % tz == 0 && g[1] % ty == 0 && g[2] % tx == 0 && mbw >= 1 && mbw <= 5 && (tz + 1) * (ty + 2) * (tx + 2) <= 320);
    EXPECT(planes ? (ty * tx <= 32 && mbw == tz) : tz * ty * tx <= 32 * mbw);
    cm::ConvArgs a{};
    a.par = 1; a.ntaps = 8; a.td = 2; a.CK = 32; a.C0 = 64; a.Co = 64; a.Zs = g[0]; a.Ys = g[1]; a.Xs = g[2];
    a.Zo = 2 * g[0]; a.Yo = 2 * g[1]; a.Xo = 2 * g[2]; a.bz = tz; a.by = ty; a.bx = tx; a.ntz = g[0] / tz; a.nty = g[1] / ty; a.ntx = g[2] / tx;
    EXPECT(cm::conv_ups_ok(a, mbw, planes, 1) && cm::conv_ups_ok(a, mbw, planes, 2) && cm::conv_ups_slots(a, mbw) <= MAX_SLOTS);
    a.C1 = 32;
    EXPECT(!cm::conv_ups_ok(a, mbw, planes, 1));           // concat sources / normalisation on load stay on the generic kernel
    a.C1 = 0; a.gn = reinterpret_cast<const float *>(&a);
    EXPECT(!cm::conv_ups_ok(a, mbw, planes, 1));
  }
  {
    int tz = 0, ty = 0, tx = 0, mbw = 0, planes = 0;
    EXPECT(cm::conv_ups_pick(4, 6, 18, &tz, &ty, &tx, &mbw, &planes) && tz == 4 && ty == 3 && tx == 9 && mbw == 4 && planes == 1);
    EXPECT(cm::conv_ups_pick(2, 3, 9, &tz, &ty, &tx, &mbw, &planes) && tz == 2 && ty == 3 && tx == 9 && mbw == 2 && planes == 1);
  }
  // the whole-sample kernel accepts the quarter resolution of the ATC / CR-120 grids and refuses what it cannot stage
  cm::QrArgs q{};
  q.C0 = 128; q.Co = 128; q.groups = 8; q.Y = 3; q.X = 9;
  EXPECT(cm::conv_qr_ok(q));
  q.Y = 7; q.X = 6; q.C1 = 128;
  EXPECT(cm::conv_qr_ok(q));
  q.Y = 6; q.X = 18;
  EXPECT(!cm::conv_qr_ok(q));                       // 108 voxels per plane
  q.Y = 3; q.X = 9; q.Co = 48;
  EXPECT(!cm::conv_qr_ok(q));
}

void test_misc_errors() {
  EXPECT(cm_abi_version() == CM_ABI_VERSION);
  EXPECT(cm_device_count(nullptr) != 0);
  EXPECT(cm_model_destroy(nullptr) == 0);
  EXPECT(cm_schedule_destroy(nullptr) == 0);
  EXPECT(cm_train_set_lr(nullptr, 1.f) != 0);
  EXPECT(cm_train_set_sample_base(nullptr, 0) != 0);
  int32_t n = 0;
  EXPECT(cm_model_num_params(nullptr, &n) != 0);
  EXPECT(cm_profile_enable(nullptr, 1) != 0);
  EXPECT(cm_frame_metrics(0, nullptr, nullptr, 1, 1, 1, 1, 1, nullptr, nullptr) != 0);
  EXPECT(std::string(cm_last_error()).size() > 0);
  // linspace restatement: endpoints and symmetry
  EXPECT(linspace_f32(0.f, 1.f, 10, 0) == 0.f && linspace_f32(0.f, 1.f, 10, 9) == 1.f && linspace_f32(3.f, 5.f, 1, 0) == 3.f);
}

}  // namespace

int main() {
  test_schedule();
  test_plan_and_params();
  test_packers();
  test_tile_planner();
  test_round3_packers();
  test_misc_errors();
  printf("selftest ok: %d checks\n", g_checks);
  return 0;
}
