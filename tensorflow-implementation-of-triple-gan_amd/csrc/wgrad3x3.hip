// Filter gradient of a 3x3 / stride 1 / SAME convolution with the activation tile read ONCE for the nine taps (round 2): the fast path
// behind tg_wgrad_f32 / tg_wgrad_bf16 for the classifier's 3x3 layers.
//
//   slab[split][tap][c][n] = sum over the pixels p of the split of  x[p + tap][c] * dy[p][n]
//
// The generic wgrad_f32_kernel (igemm.hip) gives every (tap, channel tile, column tile, pixel split) its own workgroup, so each
// activation tile is fetched from L2 by nine workgroups and each gradient tile by nine more.  Here a workgroup owns 32 input channels x
// 128 output channels x ALL nine taps — 9 x (32 x 128) fp32 accumulators = 144 registers in each of four consumer waves (wave w:
// columns 32w .. 32w+31) — and walks the pixel tiles of its split: per tile the (R+2) x (W+2) input HALO of its 32 channels and the
// R x W tile of the gradient's 128 columns go to LDS once, and the taps read the halo at shifted pixel rows.
//   * roles as in conv3x3_bf16.hip: waves 0-3 multiply, waves 4-7 fetch the next tile (global -> registers -> LDS, two LDS stages, one
//     barrier per tile);
//   * exact fp32 (v_mfma_f32_32x32x2_f32, 64 pixels per tile): D[c][n] += x^T[c][k] dy[k][n] over pixel pairs k; lane half h of both
//     operands takes pixel 2s + h, so A is one ds_read_b32 of a 128-byte halo row per tap (lane = channel) and B one of a 512-byte
//     gradient row (lane = column): ten dword reads per nine MFMAs, all conflict-free;
//   * bf16 (v_mfma_f32_32x32x16_bf16, 128 pixels per tile): operands rounded to bf16 once on the way into LDS; both fragments are
//     pixel-major (K-major), read with the transposing ds_read_b64_tr_b16 — four pixel rows x 16 channels per 16-lane group, two reads
//     per fragment.  Halo image: plain 64-byte pixel rows (the four rows of a block cover all 64 banks whatever the tap shift);
//     gradient image: 256-byte rows, 16-byte chunk XOR-swizzled by ((row & 3) << 2) | ((row >> 2) & 3) as in wgrad_f32_kernel's bf16 form.
// Numerics: as the generic kernels — fp32 accumulation, operands exact (fp32) or RNE-rounded (bf16); only the order of the pixel sum
// differs (and the pixel partition behind the `n_split` slabs, which tg_slab_reduce_f32 adds up anyway).
#include <cstdlib>
#include "tg_common.h"
#include "tg_device.h"
#include "tg_conv3x3_bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int CC = 32, NT = 128;                 // input channels / output channels per workgroup
constexpr uint32_t OOB = 0x80000000u;            // byte offset beyond any (< 2 GiB) tensor: buffer loads return 0

struct WParams {
  const float* x;
  const float* dy;
  float* slab;
  int n_img, h, ld_in, ld_out, c_out;
  int n_split, n_cc, n_nt, tiles_total, tiles_per_split;
  int tap_of[9];                                 // descriptor tap index of window position (ky, kx) = (dy + 1, dx + 1), g = 3 ky + kx
  uint32_t x_bytes, dy_bytes;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

__device__ __forceinline__ u32x2 pack4(u32x4 v) {              // 4 fp32 -> 4 bf16 (RNE), channel order kept
  return __builtin_bit_cast(u32x2, __builtin_convertvector(__builtin_bit_cast(f32x4, v), bf16x4));
}

__device__ __forceinline__ int dy_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

template <int N>
__device__ __forceinline__ void barrier_keep() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory"); }

template <int W, bool BF16>
__global__ void __launch_bounds__(512, 2) wgrad3x3_kernel(WParams p) {
  constexpr int BMW = BF16 ? 128 : 64;                           // pixels per tile
  constexpr int R = BMW / W, HW_ = W + 2, HP = (R + 2) * HW_;
  constexpr int XROW = BF16 ? 64 : 128, DROW = BF16 ? 256 : 512; // bytes per halo pixel (32 channels) / per gradient pixel (128 columns)
  constexpr int X_BYTES = (HP * XROW + 255) / 256 * 256, D_BYTES = BMW * DROW, STAGE = X_BYTES + D_BYTES;
  constexpr int X_IT = (HP * 8 + 255) / 256, D_IT = BMW * 32 / 256;   // 16-byte global loads per loader thread and tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  // consecutive logical workgroups = the (channel chunk, column tile) pairs of ONE pixel split: they stream the same tiles at the same
  // pace and meet in one XCD's L2
  int l = xcd_remap(blockIdx.x, gridDim.x);
  const int cc = l % p.n_cc;
  l /= p.n_cc;
  const int nt = l % p.n_nt;
  const int sp = l / p.n_nt;
  const int c0 = cc * CC, n0 = nt * NT;
  const int t0 = sp * p.tiles_per_split;
  const int t1 = min(p.tiles_total, t0 + p.tiles_per_split);
  const int tiles_per_img = p.h * W / BMW;

  if (wave >= 4) {
    // ================================================= loaders =====================================================================
    if (t0 >= t1) return;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);
    const int lt = tid - 256;
    // Two tiles of loads are in flight (register sets A and B, the loop is unrolled by two so that both are statically named): the loads
    // of tile t + 2 are issued at the top of iteration t, before the set holding tile t + 1 is converted and written to the other LDS
    // stage, and stay in flight across the barrier (barrier_keep<N>) — a load has a whole iteration to land.  With one set, issued and
    // consumed in the same iteration, the loaders spent an HBM latency per 2 304-cycle step (matrix pipe busy 27-35 %, PMC).
    // Addressing is kept out of the registers: unit u = lt + 256 i of a tile is pixel (lt >> 5) + 8 i of the gradient tile (its 16-byte
    // unit lt & 31) — one per-lane offset + a scalar per i — and halo pixel (lt >> 3) + 32 i (unit lt & 7).
    constexpr int NLD = X_IT + D_IT;
    const int dq = lt & 31, dpx = lt >> 5, xq = lt & 7, xhp = lt >> 3;
    const uint32_t d_voff = (uint32_t)(dpx * p.ld_out + n0 + 4 * dq) * 4u;
    const uint32_t d_step = (uint32_t)(8 * p.ld_out * 4);
    auto x_voff = [&](int t, int i) -> uint32_t {                  // halo pixel xhp + 32 i of tile t: global byte offset, OOB outside the image
      const int img = t / tiles_per_img, row0 = (t - img * tiles_per_img) * R;
      const int hp = xhp + 32 * i;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int iy = row0 + hy - 1, ix = hx - 1;
      const bool ok = hp < HP && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)W;
      return ok ? (uint32_t)(((img * p.h + iy) * W + ix) * p.ld_in + c0 + 4 * xq) * 4u : OOB;
    };
    struct Set { u32x4 x[X_IT], d[D_IT]; };
    auto gload = [&](Set& r, int t) {
      const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(t * BMW * p.ld_out * 4);
#pragma unroll
      for (int i = 0; i < X_IT; ++i) r.x[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, x_voff(t, i), 0, 0);
#pragma unroll
      for (int i = 0; i < D_IT; ++i) r.d[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_d, d_voff, so + (uint32_t)i * d_step, 0);
    };
    auto sstore = [&](const Set& r, int buf) {
      unsigned char* xs = smem + buf * STAGE;
      unsigned char* ds = xs + X_BYTES;
#pragma unroll
      for (int i = 0; i < X_IT; ++i)
        if (i + 1 < X_IT || xhp + 32 * i < HP) {
          unsigned char* dst = xs + (xhp + 32 * i) * XROW + xq * (BF16 ? 8 : 16);
          if constexpr (BF16) *reinterpret_cast<u32x2*>(dst) = pack4(r.x[i]);
          else *reinterpret_cast<u32x4*>(dst) = r.x[i];
        }
#pragma unroll
      for (int i = 0; i < D_IT; ++i) {
        const int px = dpx + 8 * i;
        if constexpr (BF16) *reinterpret_cast<u32x2*>(ds + dy_off(px, dq >> 1) + 8 * (dq & 1)) = pack4(r.d[i]);
        else *reinterpret_cast<u32x4*>(ds + px * DROW + dq * 16) = r.d[i];
      }
    };
    Set A, B;
    gload(A, t0);
    if (t0 + 1 < t1) gload(B, t0 + 1);
    sstore(A, 0);                                                 // waits for set A only (the compiler counts B's younger loads out)
    if (t0 + 1 < t1) barrier_keep<NLD>();                         // (P) the first tile is in LDS; B stays in flight
    else barrier_keep<0>();
    for (int t = t0; t < t1; t += 2) {
      // iteration t (consumers on stage 0): tile t + 1 is in set B, tile t + 2 goes to set A
      const bool more2 = t + 2 < t1;
      if (more2) gload(A, t + 2);
      if (t + 1 < t1) sstore(B, 1);
      if (more2) barrier_keep<NLD>();                             // (S)
      else barrier_keep<0>();
      if (t + 1 >= t1) break;
      // iteration t + 1 (consumers on stage 1): tile t + 2 is in set A, tile t + 3 goes to set B
      const bool more3 = t + 3 < t1;
      if (more3) gload(B, t + 3);
      if (more2) sstore(A, 0);
      if (more3) barrier_keep<NLD>();                             // (S)
      else barrier_keep<0>();
    }
  } else {
    // ================================================= consumers ===================================================================
    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    if (t0 < t1) {
      __syncthreads();                                            // (P)
      int buf = 0;
      for (int t = t0; t < t1; ++t) {
        const unsigned char* xs = smem + buf * STAGE;
        const unsigned char* ds = xs + X_BYTES;
        // Accumulator g = (ky, kx) in window order; every LDS address below is ONE per-lane base + a compile-time offset (the loops are
        // fully unrolled and the taps enumerated geometrically — the descriptor's tap order only decides which slab a tap is stored to),
        // so the K loop holds no address arithmetic: on gfx950 the fp32 MFMA shares the vector ALU's issue slots, every
        // VALU instruction in this loop is matrix time lost (measured: 96 TFLOP/s with per-tap v_mad address updates).
        if constexpr (BF16) {
          // 16 consecutive pixels of one image row per step; lane (i = lane & 31, k-group h): pixels 8h .. 8h+7 of channel / column i
          const int q = (lane >> 2) & 3, cq = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
          const int cs = 32 * wave + cq;
          const unsigned char* xa = xs + (8 * half + q) * XROW + cq * 2;
          // gradient rows 16ks + 8h + q (+4): row & 3 = q and (row >> 2) & 3 = 2h (+1) for every ks, so the swizzled chunk is a per-lane constant
          const unsigned char* dlo = ds + dy_off(8 * half + q, cs >> 3) + 8 * ((cs >> 2) & 1);
          const unsigned char* dhi = ds + dy_off(8 * half + q + 4, cs >> 3) + 8 * ((cs >> 2) & 1);
#pragma unroll
          for (int ks = 0; ks < BMW / 16; ++ks) {
            const int p0 = 16 * ks, ty = p0 / W, tx0 = p0 - ty * W;
            const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dlo + p0 * DROW));
            const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(dhi + p0 * DROW));
            const bf16x8 b = __builtin_bit_cast(bf16x8, __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int g = 0; g < 9; ++g) {
              const int hp_c = (ty + g / 3) * HW_ + tx0 + g % 3;  // halo pixel of the step's first pixel under tap g (compile-time)
              const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xa + hp_c * XROW));
              const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xa + (hp_c + 4) * XROW));
              const bf16x8 a = __builtin_bit_cast(bf16x8, __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7));
              acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[g], 0, 0, 0);       // D[channel][column]
            }
          }
        } else {
          // one pixel pair of one image row per step; lane half h takes pixel 2s + h
          const unsigned char* xa = xs + half * XROW + col * 4;
          const unsigned char* db = ds + half * DROW + (32 * wave + col) * 4;
#pragma unroll
          for (int ks = 0; ks < BMW / 2; ++ks) {
            const int p0 = 2 * ks, ty = p0 / W, tx0 = p0 - ty * W;
            const float b = *reinterpret_cast<const float*>(db + p0 * DROW);
#pragma unroll
            for (int g = 0; g < 9; ++g) {
              const int hp_c = (ty + g / 3) * HW_ + tx0 + g % 3;
              const float a = *reinterpret_cast<const float*>(xa + hp_c * XROW);
              acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g], 0, 0, 0);
            }
          }
        }
        __syncthreads();                                          // (S)
        buf ^= 1;
      }
    }
    // slab[split][tap][c][n]: row c = (r & 3) + 8*(r >> 2) + 4*half of D, lane = column -> each half-wave stores one 128-byte line
#pragma unroll
    for (int g = 0; g < 9; ++g) {
      float* out = p.slab + ((int64_t)(sp * 9 + p.tap_of[g]) * p.ld_in + c0) * p.c_out + n0 + 32 * wave + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(int64_t)((r & 3) + 8 * (r >> 2) + 4 * half) * p.c_out] = acc[g][r];
    }
  }
}

template <int W>
void launch(const WParams& p, bool bf16, hipStream_t s) {
  const dim3 grid(p.n_cc * p.n_nt * p.n_split);
  if (bf16) hipLaunchKernelGGL((wgrad3x3_kernel<W, true>), grid, dim3(512), 0, s, p);
  else hipLaunchKernelGGL((wgrad3x3_kernel<W, false>), grid, dim3(512), 0, s, p);
}

const bool g_off = getenv("TG_NO_WGRAD3X3") != nullptr;          // A/B switches, read once at library load
const bool g_no_f32 = getenv("TG_NO_WGRAD3X3_F32") != nullptr;   // fp32 launches stay on the generic kernel

}  // namespace

namespace tg {

// 3x3 window (each tap once), stride 1, output grid = input grid, width 16 / 32 / 64, whole tiles of image rows, 32 | ld_in, 128 | c_out;
// with the default policy (tg_conv3x3_policy 0) only when the launch puts a workgroup on 60 % ... 200 % of the compute units (a caller
// that passes the split of tg_wgrad_splits[_bf16] does)
bool wgrad3x3_applicable(const tg_igemm_desc* d, int n_split, bool bf16, int policy, int compute_units) {
  if (g_off || policy == 2) return false;
  if (d->n_taps != 9 || d->n_group != 0) return false;
  if (d->s_y != 1 || d->s_x != 1 || d->os_y != 1 || d->os_x != 1 || d->oo_y != 0 || d->oo_x != 0) return false;
  if (d->h_v != d->h_in || d->w_v != d->w_in || d->h_out != d->h_in || d->w_out != d->w_in) return false;
  if (d->w_in != 16 && d->w_in != 32 && d->w_in != 64) return false;
  const int bmw = bf16 ? 128 : 64;
  if ((d->h_in * d->w_in) % bmw || bmw % d->w_in) return false;
  if (d->ld_in % CC || d->c_out % NT || d->c_out > d->ld_out) return false;
  bool seen[9] = {false, false, false, false, false, false, false, false, false};
  for (int t = 0; t < 9; ++t) {
    if (d->dy[t] < -1 || d->dy[t] > 1 || d->dx[t] < -1 || d->dx[t] > 1) return false;
    const int k = (d->dy[t] + 1) * 3 + d->dx[t] + 1;
    if (seen[k]) return false;
    seen[k] = true;
  }
  if (policy == 0) {
    // exact fp32: 130 / 128 / 139 TFLOP/s on conv1_2 / conv2_1 / conv2_2 with one workgroup on every CU (the split rule below) against
    // the generic kernel's 125 / 118 / 129; TG_NO_WGRAD3X3_F32 switches it off for A/B runs
    if (!bf16 && g_no_f32) return false;
    const long wgs = (long)(d->ld_in / CC) * (d->c_out / NT) * n_split;
    if (wgs * 10 < (long)compute_units * 6 || wgs > 2L * compute_units) return false;
  }
  return true;
}

// pixel splits that put one workgroup of this kernel on every compute unit (tg_wgrad_splits / tg_wgrad_splits_bf16); 0: the layer is not this kernel's
int wgrad3x3_splits(const tg_igemm_desc* d, bool bf16, int policy, int compute_units) {
  const long per_split = (long)(d->ld_in / CC) * (d->c_out / NT);
  if (per_split < 1) return 0;
  const long tiles = (long)d->n_img * d->h_in * d->w_in / (bf16 ? 128 : 64);
  long ns = compute_units / per_split;
  if (ns > tiles / 2) ns = tiles / 2;                            // at least two tiles per split
  if (ns < 1) ns = 1;
  return wgrad3x3_applicable(d, (int)ns, bf16, policy, compute_units) ? (int)ns : 0;
}

int wgrad3x3_launch(const tg_igemm_desc* d, const float* in, const float* dout, float* slab, int n_split, uint32_t in_bytes, uint32_t dout_bytes,
                    hipStream_t s, bool bf16) {
  WParams p;
  p.x = in; p.dy = dout; p.slab = slab;
  p.n_img = d->n_img; p.h = d->h_in; p.ld_in = d->ld_in; p.ld_out = d->ld_out; p.c_out = d->c_out;
  p.n_split = n_split; p.n_cc = d->ld_in / CC; p.n_nt = d->c_out / NT;
  const int bmw = bf16 ? 128 : 64;
  p.tiles_total = d->n_img * d->h_in * d->w_in / bmw;
  p.tiles_per_split = (p.tiles_total + n_split - 1) / n_split;
  for (int t = 0; t < 9; ++t) p.tap_of[(d->dy[t] + 1) * 3 + d->dx[t] + 1] = t;
  p.x_bytes = in_bytes; p.dy_bytes = dout_bytes;
  if (d->w_in == 16) launch<16>(p, bf16, s);
  else if (d->w_in == 32) launch<32>(p, bf16, s);
  else launch<64>(p, bf16, s);
  TG_CHECK_LAUNCH("wgrad3x3_kernel");
  return TG_OK;
}

}  // namespace tg
