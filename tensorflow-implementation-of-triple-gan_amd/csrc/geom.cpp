// Host-side geometry of the implicit-GEMM family: every conv-like op of the hot path as a tg_igemm_desc, the pixel split and
// slab size of the filter-gradient launch, and the tile pick — ONE copy of each rule, behind the C ABI, so that a host binding in
// any language needs no geometry code of its own (include/tg_kernels.h "descriptor builders").  No device work here.
//
// Padding arithmetic is TensorFlow's: SAME: out = ceil(in / s), total = max((out-1)*s + k - in, 0),
// before = total / 2 (the extra pixel goes after); VALID: out = (in - k) / s + 1, no padding.  conv2d_transpose 'same' with stride s:
// out[s*i + k - before] += in[i] * W[k].
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "tg_common.h"
#include "tg_geom.h"
#include "tg_conv3x3_bf16.h"

namespace {

inline int pad32(int c) { return (c + 31) / 32 * 32; }

struct Pad { int out, before; };

Pad same_pad(int n, int k, int s) {
  const int out = (n + s - 1) / s;
  int total = (out - 1) * s + k - n;
  if (total < 0) total = 0;
  return {out, total / 2};
}

Pad out_size(int n, int k, int s, int pad_same) { return pad_same ? same_pad(n, k, s) : Pad{(n - k) / s + 1, 0}; }

struct Tap { int dy, dx, tw; };

int fill(tg_igemm_desc* d, int n_img, int h_in, int w_in, int ld_in, int h_v, int w_v, int s, int h_out, int w_out, int ld_out, int os, int oo_y,
         int oo_x, int c_out, int n_store, const Tap* taps, int n_taps, int64_t w_sn, int64_t w_st, int act, float alpha) {
  TG_REQUIRE(n_taps >= 1 && n_taps <= TG_MAX_TAPS, "geom: %d taps (at most %d)", n_taps, TG_MAX_TAPS);
  std::memset(d, 0, sizeof *d);
  d->n_img = n_img; d->h_in = h_in; d->w_in = w_in; d->ld_in = ld_in;
  d->h_v = h_v; d->w_v = w_v; d->s_y = s; d->s_x = s;
  d->h_out = h_out; d->w_out = w_out; d->ld_out = ld_out;
  d->os_y = os; d->os_x = os; d->oo_y = oo_y; d->oo_x = oo_x;
  d->c_out = c_out; d->n_store = n_store; d->n_taps = n_taps;
  for (int i = 0; i < n_taps; ++i) {
    TG_REQUIRE(taps[i].dy >= -128 && taps[i].dy < 128 && taps[i].dx >= -128 && taps[i].dx < 128, "geom: tap offset out of the int8 range");
    d->dy[i] = (int8_t)taps[i].dy; d->dx[i] = (int8_t)taps[i].dx; d->tapw[i] = (int16_t)taps[i].tw;
  }
  d->w_sn = w_sn; d->w_st = w_st;
  d->act = act; d->alpha = alpha;
  return TG_OK;
}

#define GEOM_ARGS_OK(cond, what) TG_REQUIRE(cond, "geom: %s", what)

}  // namespace

namespace tg {

// ---- tile pick of igemm_impl (csrc/igemm.hip) — the quantisation cost model, see the comments there ------------------------------------
static double g_eff64 = 1.02;
static int g_force_bm = 0, g_force_bn = 0;
static bool g_eff64_env = false;

void igemm_tuning_from_env() {           // read ONCE, at library load (runtime.cpp), never per launch
  if (const char* e = getenv("TG_IGEMM_EFF64")) { g_eff64 = atof(e); g_eff64_env = true; }
  if (const char* f = getenv("TG_IGEMM_TILE")) sscanf(f, "%d,%d", &g_force_bm, &g_force_bn);
}

bool igemm_pick_tile(const tg_igemm_desc* descs, int n_desc, bool colsum, const int32_t* seg_rows, int nseg, bool bf16, int* bm_out, int* bn_out) {
  struct Cand { int bm, bn; double eff; };
  const Cand cands[] = {{128, 128, 1.00}, {64, 128, 0.97}, {64, 64, g_eff64}, {128, 64, 0.95}, {32, 128, 0.85}, {128, 32, 0.70}};
  const tg_igemm_desc* d = &descs[0];
  const int64_t M = (int64_t)d->n_img * d->h_v * d->w_v;
  double taps = 0;
  int max_taps = 0;
  for (int i = 0; i < n_desc; ++i) { taps += descs[i].n_taps; max_taps = descs[i].n_taps > max_taps ? descs[i].n_taps : max_taps; }
  int bm = 0, bn = 0;
  double best = 1e300;
  // TG_IGEMM_TILE (tile audits, tools/tile_audit_step.sh): the named tile wherever it is a candidate, the model's pick elsewhere
  for (int forced = g_force_bm ? 1 : 0; forced >= 0 && best >= 1e299; --forced)
  for (const Cand& c : cands) {
    // a tile may overhang the last columns (c_out = 544 = 8.5 x 64: nine 64-column tiles instead of seventeen 32-column ones); the
    // quantisation below charges the idle columns
    if (d->c_out % c.bn && c.bn > d->c_out) continue;
    if (forced && (c.bm != g_force_bm || c.bn != g_force_bn)) continue;
    bool seg_ok = true;                                      // COLSUM: a tile may straddle at most one application boundary
    for (int i = 0; colsum && i < nseg; ++i) seg_ok = seg_ok && seg_rows[i] >= c.bm;
    if (!seg_ok) continue;
    const int64_t per_sub = ((M + c.bm - 1) / c.bm) * ((d->c_out + c.bn - 1) / c.bn);
    // K-tiles per CU: whole rounds for one problem; for the unequal sub-problems of a stride-2 launch (4/6/6/9 taps of a 5x5
    // transposed conv) the longest workgroup bounds the launch from below, which is what pushes those to small tiles
    double iters;
    if (n_desc == 1) {
      iters = (double)((per_sub + 255) / 256) * max_taps;
    } else {
      // the sub-problems are mixed over the compute units (rotating order, see the kernel): a unit's load is the mean plus about half
      // of the longest workgroup — measured on the generator's layers: 64x64 tiles 0.124 ms, 64x128 0.153 ms, equal mean load
      const double total = (double)per_sub * taps / 256.0 + 0.5 * max_taps;
      iters = total > max_taps ? total : max_taps;
    }
    // bf16 operands: the conversion work per tile favours the large tile (measured: CIFAR-10 bf16 step 8.06 ms with 0.96, 8.55 ms with 1.02)
    const double eff = (bf16 && c.bm == 64 && c.bn == 64 && !g_eff64_env) ? 0.96 : c.eff;
    const double t = iters * c.bm * c.bn / eff;
    if (t < best) { best = t; bm = c.bm; bn = c.bn; }
  }
  if (best >= 1e299) return false;
  *bm_out = bm; *bn_out = bn;
  return true;
}

void igemm_sub_order(const tg_igemm_desc* descs, int n_desc, int* order) {
  for (int i = 0; i < n_desc; ++i) order[i] = i;
  for (int i = 0; i < n_desc; ++i)
    for (int j = i + 1; j < n_desc; ++j)
      if (descs[order[j]].n_taps > descs[order[i]].n_taps) { int t_ = order[i]; order[i] = order[j]; order[j] = t_; }
}

// ---- work units: which tiles are cut along K -----------------------------------------------------------------------------------------
// Why: (a) a launch of fewer tiles than resident-workgroup slots (2 per compute unit) leaves compute units idle or with ONE workgroup,
// whose load / LDS / barrier stalls nothing hides (measured: 225 tiles, 200 K-tiles each, 0.151 ms; cut in two 0.125 ms; the 32-tile
// tail of a split 130-image classifier launch 0.058 -> 0.022 ms); (b) the 9 / 6 / 6 / 4-tap parities of a stride-2 transposed conv are
// resident together and the 9-tap workgroups finish last (0.146 -> 0.120 ms with the 9-tap tiles cut in two).  The price is the partial
// sums' round trip through scratch and a fix-up launch (~10 us end to end), so short launches and launches that fill the chip anyway are
// left alone.  MEASURED AND REJECTED (round 3, profiles/r03_split_ab.txt): cutting only the tiles of the last partial round of a launch
// with more tiles than slots (1 128 tiles of conv3: 0.231 -> 0.272 ms) — workgroups do not run in lockstep rounds, a compute unit whose
// partner slot is empty runs the remaining workgroup faster, and the fix-up launch waits for the whole main launch.
static const bool g_no_split = getenv("TG_IGEMM_NOSPLIT") != nullptr;      // A/B switch, read once at library load

void igemm_schedule(int n_sub, const int* nk, int64_t T, int bm, int bn, int slots, bool allow_split, IgemmSched* o) {
  std::memset(o, 0, sizeof *o);
  for (int s = 0; s < 4; ++s) o->ks[s] = 1;
  if (slots < 1) slots = 512;
  // one K-tile (32 deep) of a bm x bn tile on one of `slots` workgroup slots of a 157 TFLOP/s chip, in microseconds
  const double ktile_us = 2.0 * bm * bn * 32 / (157.3e12 / slots) * 1e6;
  int mink = (int)(8.0 / ktile_us + 0.999);                // a unit is at least ~8 us of matrix work
  if (mink < 2) mink = 2;
  const bool split_ok = allow_split && !g_no_split && T > 0;
  if (n_sub == 1) {
    const int K = nk[0];
    o->nfull = (int)T;
    o->n_units = (int)T;
    // under-filled launches only: all units resident at once; few tiles (a quarter of the slots) or a long reduction (>= 100 K-tiles) —
    // 200-250 tiles of 27-45 K-tiles measured neutral to -15 %
    if (!split_ok || T * 2 > slots || !(T * 4 <= slots || K >= 100)) return;
    int ks = (int)(slots / T);
    if (ks > 8) ks = 8;
    while (ks > 1 && K / ks < mink) --ks;
    if (ks < 2) return;
    o->ks[0] = ks;
    o->nfull = 0;
    o->n_fix = (int)T;
    o->n_units = (int)(T * ks);
    o->ws_bytes = T * ks * (int64_t)bm * bn * 4;
    return;
  }
  // several sub-problems: every tile index owns one unit per (sub-problem, K segment).  Cut only while everything stays resident
  // (T * sum ks <= slots): bring the long sub-problems' units down towards the shortest one's length.
  int best_ks[4] = {1, 1, 1, 1};
  if (split_ok && T * n_sub <= slots) {
    int shortest = nk[0];
    for (int s = 1; s < n_sub; ++s) shortest = nk[s] < shortest ? nk[s] : shortest;
    for (;;) {
      int longest = 0, ls = -1, sum = 0;
      for (int s = 0; s < n_sub; ++s) {
        sum += best_ks[s];
        const int len = (nk[s] + best_ks[s] - 1) / best_ks[s];
        if (len > longest) { longest = len; ls = s; }
      }
      if (ls < 0 || best_ks[ls] >= 4 || T * (sum + 1) > slots || sum + 1 > 16) break;
      if (nk[ls] / (best_ks[ls] + 1) < mink || longest * 4 <= shortest * 5) break;      // already within 25 % of the shortest
      ++best_ks[ls];
    }
  }
  int len = 0, nslot = 0, nsp = 0;
  for (int s = 0; s < n_sub; ++s) {
    o->ks[s] = best_ks[s];
    if (best_ks[s] > 1) { o->split_sub[nsp] = (int8_t)s; o->first_slot[nsp] = (int8_t)nslot; ++nsp; }
    for (int k = 0; k < best_ks[s]; ++k) {
      o->pat_sub[len] = (int8_t)s; o->pat_k[len] = (int8_t)k;
      o->pat_slot[len] = best_ks[s] > 1 ? (int8_t)nslot++ : (int8_t)-1;
      ++len;
    }
  }
  o->pat_len = len;
  o->n_pat_split = nslot;
  o->n_split_sub = nsp;
  o->nfull = (int)T;
  o->n_units = (int)(T * len);
  o->n_fix = (int)(T * nsp);
  o->ws_bytes = T * nslot * (int64_t)bm * bn * 4;
}

static const int g_tuning_loaded = (igemm_tuning_from_env(), 0);      // at library load: never a getenv on the launch path

// widest tile that divides the dimension; odd multiples of 32 from 160 on (288 = 256 + 32 label channels, 544, 160) take 64-wide tiles with
// an overhanging last one instead of 32-wide ones
int wgrad_tile(int n) { return n % 128 == 0 ? 128 : ((n % 64 == 0 || n >= 160) ? 64 : 32); }

}  // namespace tg

extern "C" {

int tg_conv2d_desc_fwd(int n, int h, int w, int ld_in, int c_out, int k, int stride, int pad_same, int ld_out, int n_store, int act, float alpha,
                       tg_igemm_desc* d) {
  GEOM_ARGS_OK(d && n > 0 && h > 0 && w > 0 && k >= 1 && k <= 5 && stride >= 1 && (pad_same || (h >= k && w >= k)), "conv2d_desc_fwd: bad arguments");
  const Pad py = out_size(h, k, stride, pad_same), px = out_size(w, k, stride, pad_same);
  Tap taps[TG_MAX_TAPS];
  int nt = 0;
  for (int ky = 0; ky < k; ++ky)
    for (int kx = 0; kx < k; ++kx) taps[nt++] = {ky - py.before, kx - px.before, ky * k + kx};
  if (ld_out <= 0) ld_out = c_out;
  if (n_store < 0) n_store = c_out;
  return fill(d, n, h, w, ld_in, py.out, px.out, stride, py.out, px.out, ld_out, 1, 0, 0, c_out, n_store, taps, nt, (int64_t)k * k * ld_in, ld_in, act, alpha);
}

int tg_conv2d_desc_dgrad(int n, int h, int w, int c_in_pad, int ld_dy, int k, int stride, int pad_same, int ld_out, int n_store, tg_igemm_desc* descs,
                         int32_t* n_desc_out) {
  GEOM_ARGS_OK(descs && n_desc_out && n > 0 && h > 0 && w > 0 && k >= 1 && k <= 5 && stride >= 1 && stride <= 2, "conv2d_desc_dgrad: bad arguments");
  const Pad py = out_size(h, k, stride, pad_same), px = out_size(w, k, stride, pad_same);
  if (ld_out <= 0) ld_out = c_in_pad;
  if (n_store < 0) n_store = c_in_pad;
  int nd = 0;
  for (int qy = 0; qy < stride; ++qy)
    for (int qx = 0; qx < stride; ++qx) {
      Tap taps[TG_MAX_TAPS];
      int nt = 0;
      for (int ky = 0; ky < k; ++ky) {
        if (((qy + py.before - ky) % stride + stride) % stride) continue;
        for (int kx = 0; kx < k; ++kx) {
          if (((qx + px.before - kx) % stride + stride) % stride) continue;
          // floor division (Python's //): the numerators are exact multiples of the stride here
          taps[nt++] = {(qy + py.before - ky) / stride, (qx + px.before - kx) / stride, ky * k + kx};
        }
      }
      const int hv = (h - qy + stride - 1) / stride, wv = (w - qx + stride - 1) / stride;
      if (nt == 0 || hv <= 0 || wv <= 0) continue;          // caller must zero-fill such outputs (does not occur for the nets here)
      int rc = fill(&descs[nd++], n, py.out, px.out, ld_dy, hv, wv, 1, h, w, ld_out, stride, qy, qx, c_in_pad, n_store, taps, nt, ld_dy,
                    (int64_t)c_in_pad * ld_dy, TG_ACT_NONE, 0.2f);
      if (rc != TG_OK) return rc;
    }
  *n_desc_out = nd;
  return TG_OK;
}

int tg_conv2d_desc_wgrad(int n, int h, int w, int ld_in, int c_out_pad, int k, int stride, int pad_same, int ld_dy, tg_igemm_desc* d) {
  return tg_conv2d_desc_fwd(n, h, w, ld_in, c_out_pad, k, stride, pad_same, ld_dy <= 0 ? c_out_pad : ld_dy, c_out_pad, TG_ACT_NONE, 0.2f, d);
}

int tg_deconv5x5s2_desc_fwd(int n, int h, int w, int ld_in, int c_out_pad, int ld_out, int n_store, int act, tg_igemm_desc* descs, int32_t* n_desc_out) {
  GEOM_ARGS_OK(descs && n_desc_out && n > 0 && h > 0 && w > 0, "deconv5x5s2_desc_fwd: bad arguments");
  const int k = 5, stride = 2;
  const int pt = same_pad(h * stride, k, stride).before, pl = same_pad(w * stride, k, stride).before;
  if (ld_out <= 0) ld_out = c_out_pad;
  if (n_store < 0) n_store = c_out_pad;
  int nd = 0;
  for (int qy = 0; qy < stride; ++qy)
    for (int qx = 0; qx < stride; ++qx) {
      Tap taps[TG_MAX_TAPS];
      int nt = 0;
      for (int ky = 0; ky < k; ++ky) {
        if ((qy + pt - ky) % stride) continue;
        for (int kx = 0; kx < k; ++kx)
          if ((qx + pl - kx) % stride == 0) taps[nt++] = {(qy + pt - ky) / stride, (qx + pl - kx) / stride, ky * k + kx};
      }
      int rc = fill(&descs[nd++], n, h, w, ld_in, h, w, 1, h * stride, w * stride, ld_out, stride, qy, qx, c_out_pad, n_store, taps, nt, ld_in,
                    (int64_t)c_out_pad * ld_in, act, 0.2f);
      if (rc != TG_OK) return rc;
    }
  *n_desc_out = nd;
  return TG_OK;
}

int tg_deconv5x5s2_desc_fwd_merged(int n, int h, int w, int ld_in, int c_out, int ld_out, int n_store, int act, tg_igemm_desc* d, int32_t* n_group_out,
                                   int32_t* tapmap) {
  GEOM_ARGS_OK(d && n_group_out && tapmap && n > 0 && h > 0 && w > 0 && c_out > 0, "deconv5x5s2_desc_fwd_merged: bad arguments");
  const int k = 5, stride = 2;
  const int pt = same_pad(h * stride, k, stride).before, pl = same_pad(w * stride, k, stride).before;
  const int n_group = c_out;                               // channels per parity group
  const int n_pad = pad32(4 * n_group);
  for (int i = 0; i < 36; ++i) tapmap[i] = -1;
  for (int qy = 0; qy < stride; ++qy)
    for (int qx = 0; qx < stride; ++qx) {
      const int g = qy * stride + qx;
      for (int ky = 0; ky < k; ++ky) {
        if ((qy + pt - ky) % stride) continue;
        for (int kx = 0; kx < k; ++kx) {
          if ((qx + pl - kx) % stride) continue;
          const int dy = (qy + pt - ky) / stride, dx = (qx + pl - kx) / stride;
          GEOM_ARGS_OK(dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1, "deconv5x5s2_desc_fwd_merged: tap outside the 3x3 window");
          tapmap[g * 9 + (dy + 1) * 3 + (dx + 1)] = ky * k + kx;
        }
      }
    }
  Tap taps[9];
  int nt = 0;
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) taps[nt++] = {dy, dx, (dy + 1) * 3 + (dx + 1)};
  if (n_store < 0) n_store = c_out;
  int rc = fill(d, n, h, w, ld_in, h, w, 1, h * stride, w * stride, ld_out, stride, 0, 0, n_pad, n_store, taps, nt, (int64_t)9 * ld_in, ld_in, act, 0.2f);
  if (rc != TG_OK) return rc;
  d->n_group = n_group;
  *n_group_out = n_group;
  return TG_OK;
}

int tg_deconv5x5s2_desc_dgrad(int n, int h, int w, int c_in_pad, int ld_dy, int ld_out, int n_store, tg_igemm_desc* d) {
  GEOM_ARGS_OK(d && n > 0 && h > 0 && w > 0, "deconv5x5s2_desc_dgrad: bad arguments");
  const int k = 5, stride = 2;
  const int pt = same_pad(h * stride, k, stride).before, pl = same_pad(w * stride, k, stride).before;
  Tap taps[TG_MAX_TAPS];
  int nt = 0;
  for (int ky = 0; ky < k; ++ky)
    for (int kx = 0; kx < k; ++kx) taps[nt++] = {ky - pt, kx - pl, ky * k + kx};
  if (ld_out <= 0) ld_out = c_in_pad;
  if (n_store < 0) n_store = c_in_pad;
  return fill(d, n, h * stride, w * stride, ld_dy, h, w, stride, h, w, ld_out, 1, 0, 0, c_in_pad, n_store, taps, nt, ld_dy, (int64_t)c_in_pad * ld_dy,
              TG_ACT_NONE, 0.2f);
}

int tg_deconv5x5s2_desc_wgrad(int n, int h, int w, int ld_dy, int c_in_pad, int ld_x, tg_igemm_desc* d) {
  return tg_conv2d_desc_fwd(n, h * 2, w * 2, ld_dy, c_in_pad, 5, 2, 1, ld_x <= 0 ? c_in_pad : ld_x, c_in_pad, TG_ACT_NONE, 0.2f, d);
}

int tg_dense_desc(int m, int ld_in, int c_out, int ld_out, int n_store, int act, int64_t w_sn, tg_igemm_desc* d) {
  GEOM_ARGS_OK(d && m > 0 && ld_in > 0 && c_out > 0, "dense_desc: bad arguments");
  const Tap t{0, 0, 0};
  return fill(d, m, 1, 1, ld_in, 1, 1, 1, 1, 1, ld_out <= 0 ? c_out : ld_out, 1, 0, 0, c_out, n_store < 0 ? c_out : n_store, &t, 1, w_sn <= 0 ? ld_in : w_sn, 0,
              act, 0.2f);
}

int tg_dense_splitk_desc(int m, int k_dim, int n_out, int splits, tg_igemm_desc* descs) {
  GEOM_ARGS_OK(descs && m > 0 && splits >= 1 && splits <= 4 && k_dim % (32 * splits) == 0 && n_out > 0, "dense_splitk_desc: bad arguments");
  const int kc = k_dim / splits;
  for (int s = 0; s < splits; ++s) {
    const Tap t{0, s, s};
    int rc = fill(&descs[s], m, 1, splits, kc, 1, 1, 1, 1, splits, n_out, 1, 0, s, n_out, n_out, &t, 1, k_dim, kc, TG_ACT_NONE, 0.2f);
    if (rc != TG_OK) return rc;
  }
  return TG_OK;
}

int tg_wgrad_splits(const tg_igemm_desc* d) {
  if (!d || d->ld_in <= 0 || d->c_out <= 0 || d->n_taps <= 0) { tg::set_error("wgrad_splits: bad descriptor"); return TG_ERR_INVALID; }
  // the classifier's 3x3 / stride-1 layers run on wgrad3x3.hip: one workgroup (32 input channels x 128 output channels x nine taps) per CU
  if (const int ns3 = tg::wgrad3x3_splits(d, false, tg::halo_policy(), tg::halo_compute_units())) return ns3;
  // generic kernel: fill ONE round of the 512 resident workgroups (256 CUs x 2) as fully as possible — 576 blocks take two rounds and run at 56 % —
  // with at least 128 pixels (4 K-tiles) per split
  const int ct = tg::wgrad_tile(d->ld_in), nt = tg::wgrad_tile(d->c_out);
  const int64_t tiles = (int64_t)d->n_taps * ((d->ld_in + ct - 1) / ct) * ((d->c_out + nt - 1) / nt);
  const int64_t m = (int64_t)d->n_img * d->h_v * d->w_v;
  int64_t ns = 512 / tiles;
  if (ns > m / 128) ns = m / 128;
  return (int)(ns < 1 ? 1 : ns);
}

int tg_wgrad_splits_bf16(const tg_igemm_desc* d) {
  if (!d || d->ld_in <= 0 || d->c_out <= 0 || d->n_taps <= 0) { tg::set_error("wgrad_splits_bf16: bad descriptor"); return TG_ERR_INVALID; }
  // the classifier's 3x3 layers run on wgrad3x3.hip with bf16 operands: one workgroup (32 input channels x 128 output channels x the nine taps)
  // per compute unit; everything else as tg_wgrad_splits
  const int ns = tg::wgrad3x3_splits(d, true, tg::halo_policy(), tg::halo_compute_units());     // 128-pixel tiles (fp32: 64)
  return ns > 0 ? ns : tg_wgrad_splits(d);
}

int64_t tg_wgrad_workspace_bytes(const tg_igemm_desc* d, int n_split) {
  if (!d || n_split < 1) { tg::set_error("wgrad_workspace_bytes: bad arguments"); return TG_ERR_INVALID; }
  return (int64_t)n_split * d->n_taps * d->ld_in * d->c_out * 4;
}

int64_t tg_filter_workspace_bytes(int n_taps, int c_in_pad, int c_out_pad) {
  if (n_taps < 1 || c_in_pad < 1 || c_out_pad < 1) { tg::set_error("filter_workspace_bytes: bad arguments"); return TG_ERR_INVALID; }
  return (int64_t)n_taps * c_in_pad * c_out_pad * 4;
}

int tg_igemm_tile(const tg_igemm_desc* descs, int n_desc, const int32_t* seg_rows, int nseg, int bf16, int32_t* bm_out, int32_t* bn_out) {
  TG_REQUIRE(descs && n_desc >= 1 && n_desc <= 4 && bm_out && bn_out, "igemm_tile: bad arguments");
  int bm = 0, bn = 0;
  TG_REQUIRE(tg::igemm_pick_tile(descs, n_desc, nseg > 0, seg_rows, nseg, bf16 != 0, &bm, &bn), "igemm_tile: no tile fits c_out=%d with the given segments",
             descs[0].c_out);
  *bm_out = bm; *bn_out = bn;
  return TG_OK;
}

int tg_igemm_colsum_supported(const tg_igemm_desc* d, const int32_t* seg_rows, int nseg) {
  if (!d || !seg_rows || nseg < 1 || nseg > 8 || d->n_group != 0) return 0;
  int64_t tot = 0;
  for (int i = 0; i < nseg; ++i) { if (seg_rows[i] < 32) return 0; tot += seg_rows[i]; }
  if (tot != (int64_t)d->n_img * d->h_v * d->w_v) return 0;
  int bm, bn;
  return tg::igemm_pick_tile(d, 1, true, seg_rows, nseg, false, &bm, &bn) ? 1 : 0;
}

}  // extern "C"
