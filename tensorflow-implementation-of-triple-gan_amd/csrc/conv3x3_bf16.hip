// 3x3 / stride 1 / SAME convolution on the bf16 matrix cores of gfx950 (BASELINE.json configs[3], "bf16 MFMA conv path"):
// the fast path behind tg_igemm_bf16 / tg_igemm_colsum_bf16 / tg_igemm_actsum_bf16 for the classifier's 3x3 layers — forward AND
// input gradient (the same geometry with the transposed filter layout; the weight strides come from the descriptor).
//
// Why not the generic implicit GEMM (igemm.hip) with bf16 operands: gathering one K-tile per (tap, channel chunk) re-reads every input
// pixel nine times through L1/L2, and at the bf16 MFMA rate (1024 FLOP/clk/SIMD, 16x the fp32 one) that path — not the matrix pipe —
// bounds the kernel (round 1: 200 TFLOP/s = 8 % of peak with fp32 LDS images converted fragment by fragment).  Here
//   * a workgroup owns 256 output pixels = R whole image rows (R*W = 256, W in {16, 32, 64}) x 128 output channels;
//   * per 64-channel chunk the (R+2) x (W+2) input HALO is loaded ONCE (fp32 in HBM -> v_cvt_pk_bf16_f32 -> bf16 in LDS; pixels outside
//     the image are out-of-range buffer loads, i.e. hardware zeros) and all nine taps read their fragments from it at shifted pixel
//     addresses: operand traffic from L2 drops ~7x, the LDS image is half the size, one ds_read_b128 feeds a whole 32x32x16 fragment;
//   * the filter tile of each (tap, chunk) — 128 rows x 64 channels — is converted on its global -> LDS path the same way;
//   * LDS rows are 128 B (64 bf16); the 16-B chunk index is XOR-swizzled with bits 1..3 of the row so that every 16-lane group of a
//     ds_read_b128 (and every ds_write_b64 of the staging pass) hits 16 different bank slots;
//   * a K-step is a group of THREE taps of one chunk (3 072 matrix-pipe cycles per SIMD): the filter of the next group arrives while
//     the current one is multiplied; one barrier per group (with one tap per barrier the matrix work between two barriers was too short
//     to hide an L2 latency: 3.5 k cycles per 1 k of matrix work);
//   * ROLE SPECIALISATION: waves 0-3 are CONSUMERS — one per SIMD, 64 pixels x all 128 channels each (2 x 4 MFMA tiles: 96 MFMAs per
//     three-tap step, six ds_read_b128 per eight MFMAs) — and never touch global memory inside the K loop; waves 4-7 are LOADERS — the
//     partner wave on each SIMD — that bring in the next filter group and the next halo while the consumers multiply.  (Rounds 1-2 went
//     through a symmetric form — all eight waves load, convert, multiply in lockstep: the matrix pipe idled 3.0 k of every 7.0 k cycles —
//     and a one-tile-per-workgroup form with an LDS-staged epilogue: 585 / 597 / 809 TFLOP/s on conv1_2 / conv2_1 / conv2_2 against
//     869 / 739 / 961 here; both were deleted in round 3, the numbers are in DESIGN.md §4.)
// The exact-fp32 form (BF16 = false) is the SAME structure on v_mfma_f32_32x32x2_f32 for the fp32 training step — fp32 LDS images of 32
// channels per chunk (the same 128-B rows and swizzle), one ds_read_b128 feeding four k-steps of both operands (lane half h of k-group g
// supplies k = 8g + 4h + s in step s), no conversion.
// Numerics: every MFMA operand is rounded to bf16 (RNE) exactly as the generic tg_*_bf16 kernels do, products accumulate in fp32 —
// the results differ from those kernels only by the order of the fp32 accumulation (channel chunks outermost here).
#include <cstdlib>
#include <type_traits>
#include "tg_common.h"
#include "tg_device.h"
#include "tg_conv3x3_bf16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifdef TG_STAMP
// Diagnostic build (csrc/Makefile target libtg_stamp.so, never shipped): cycle stamps of workgroup phases, wave 0 of a few workgroups,
// read back with tg_debug_read_conv_stamps.  Stamp values go only to this buffer; no output depends on them.
__device__ unsigned long long tg_conv_stamps[8 * 64];
#define CSTAMP(i)                                                                                        \
  do {                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    if (threadIdx.x == 0 && blockIdx.x < 64) {                                                           \
      unsigned long long t_;                                                                             \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
      tg_conv_stamps[(i) * 64 + blockIdx.x] = t_;                                                        \
    }                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  } while (0)
#define LSTAMP(var)                                                                \
  do {                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");    \
    __builtin_amdgcn_sched_barrier(0);                                             \
  } while (0)
#else
#define CSTAMP(i) do {} while (0)
#define LSTAMP(var) do {} while (0)
#endif

#ifndef TG_C3_PIPE
#define TG_C3_PIPE 1                           // consumers: fragment reads software-pipelined behind the MFMAs — 1: bf16 operands only (shipped), 2: fp32 too (measured 1.5 - 2 % SLOWER: profiles/r04_c3pipe_ab.txt), 0: the compiler's order
#endif
#ifndef TG_ABL
#define TG_ABL 0                               // timing ablations of conv3x3_pipe_kernel (tools only, results then wrong): 1 no stores, 8 half the halo loads, 16 half the filter loads, 32 no MFMAs
#endif

namespace {

constexpr int BN = 128, KC = 64;               // output channels per tile, channels per chunk
constexpr uint32_t OOB = 0x80000000u;          // byte offset beyond any (< 2 GiB) tensor: buffer loads return 0, stores are dropped

struct ConvParams {
  const float* in;
  const float* w;
  const float* bias;
  float* out;
  double* colsum;
  const float* ymul;
  int ymul_act;
  float ymul_alpha;
  int nseg, seg_rows[8];
  int n_img, h, ld_in, ld_out, c_out, n_store, act;
  float alpha;
  int64_t w_sn, w_st;                          // filter element (n, tap, c) at n*w_sn + tapw[tap]*w_st + c
  int tap[9];                                  // (tapw << 16) | ((dy & 0xff) << 8) | (dx & 0xff)
  int n_tiles_m, n_tiles_n;
  uint32_t in_bytes, w_bytes, out_bytes;
  const void* wpk;                             // bf16 filter packed as consecutive LDS images (filter_pack_kernel) — conv3x3_pipe_kernel<.., BF16 = true>
  int f32;                                     // 1: exact-fp32 operands (conv3x3_pipe_kernel<..., BF16 = false>)
  int stat2;                                   // COLSUM launches: 2 = the output is a gradient dy, colsum[seg][0][c] += dy, colsum[seg][1][c] += dy * ymul (tg_igemm_bnbwdstat_*); 1 = statistics of the ACTIVATED output act(acc + bias): colsum[seg][0][c] += v, colsum[seg][1][c] += v*v (batch norm behind the layer)
};

int compute_units() {
  static const int n = ([] { int dev = 0; hipDeviceProp_t pr; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; })();
  return n;
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack2(uint32_t a, uint32_t b) {   // 2 fp32 -> 2 bf16 (RNE): one v_cvt_pk_bf16_f32
  const f32x2 f = {__builtin_bit_cast(float, a), __builtin_bit_cast(float, b)};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
}

__device__ __forceinline__ u32x2 pack4(u32x4 v) {             // 4 fp32 -> 4 bf16 (RNE), 8 bytes, channel order kept
  u32x2 r;
  r.x = pack2(v.x, v.y);
  r.y = pack2(v.z, v.w);
  return r;
}

// v + (v of the lane the DPP control names); lanes the row mask excludes add 0
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}

// LDS byte offset of 16-B chunk `chunk` (8 bf16) of row `row`: 128-B rows, chunk index XOR-swizzled with bits 1..3 of the row.
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// fp32 filter -> bf16, laid out as the consecutive LDS images the pipelined kernel's steps consume: image (nt, chunk, tap index k9 in the
// descriptor's tap order) = 128 filter rows x 64 channels = 16 KB with the 16-byte chunks XOR-swizzled exactly as lds_off() places them,
// so that a step's three images (48 KB) go global -> LDS by LDS-DMA as they stand (1 KB per wave-instruction, no registers, no
// conversion, no ds_write) and every workgroup fetches half the bytes from L2.  One thread per 16-byte chunk; ~150 k - 600 k elements.
__global__ void __launch_bounds__(256) filter_pack_kernel(ConvParams p, int nchunks, int total) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= total) return;
  const int q = u & 7, r = (u >> 3) & 127, img = u >> 10;
  const int k9 = img % 9, c = (img / 9) % nchunks, nt = img / (9 * nchunks);
  const float* src = p.w + (int64_t)(nt * BN + r) * p.w_sn + (int64_t)(p.tap[k9] >> 16) * p.w_st + c * KC + 8 * q;
  const u32x4 lo = *reinterpret_cast<const u32x4*>(src), hi = *reinterpret_cast<const u32x4*>(src + 4);
  const u32x2 a = pack4(lo), b = pack4(hi);
  u32x4 o;
  o.x = a.x; o.y = a.y; o.z = b.x; o.w = b.y;
  *reinterpret_cast<u32x4*>(static_cast<unsigned char*>(const_cast<void*>(p.wpk)) + (size_t)img * (BN * 128) + lds_off(r, q)) = o;
}

// s_waitcnt vmcnt(N) lgkmcnt(0) + s_barrier: the N youngest vector-memory operations of this wave stay in flight across the barrier
template <int N>
__device__ __forceinline__ void barrier_keep() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory"); }

// ---------------------------------------------------------------------------------------------------------------------------------
// Persistent, tile-pipelined, role-specialised kernel.  Stamped on its one-tile-per-workgroup predecessor (conv1_2, bf16 operands): of
// a workgroup's 59 k cycles 13 k were the prologue (first halo + filter group: HBM / L2 latency with the matrix pipe idle), 29 k the K
// loop and 17 k the epilogue (128 KB staged through LDS and stored by all eight waves) — one workgroup per CU, so nothing else ran
// on the CU meanwhile, and all 256 workgroups of a round were in the same phase at once (HBM idle during the K loops, matrix pipes idle
// during loads and stores).  Here a workgroup stays resident and walks its share of the tiles:
//   * the loaders treat (tile, chunk, tap group) as ONE stream — the first halo and filter group of the next tile are fetched during the
//     last steps of the current one, so only a workgroup's first tile pays a prologue;
//   * the consumers store their accumulators straight from registers (lane = channel: each half-wave writes one 128-B line of a pixel;
//     bias / activation, the activation-gradient multiplier and the column sums are per lane) — no LDS staging, no barrier, and the
//     stores drain while the next tile's MFMAs run;
//   * column sums: per wave in registers, one LDS slot per consumer wave, summed and added to the global accumulator (one double
//     atomic per channel and tile, as before) by the loader waves after the next barrier.
// Tile order: XCD x (= blockIdx & 7) owns a contiguous eighth of the tiles (neighbouring tiles share halo rows and all share the filter
// in that XCD's L2); its 32 workgroups stride through it.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int W, bool COLSUM, bool BF16, bool STAT2 = false>
__global__ void __launch_bounds__(512, 2) conv3x3_pipe_kernel(ConvParams p) {
  static_assert(!STAT2 || COLSUM, "STAT2 is a COLSUM mode");
  constexpr int BM = 256, TPS = 3, NG = 3;
  constexpr int KCH = BF16 ? 64 : 32;
  constexpr int UPR = BF16 ? 16 : 8;
  constexpr int RPP = 256 / UPR;
  constexpr int R = BM / W, HW_ = W + 2, HP = (R + 2) * HW_;
  constexpr int A_BYTES = (HP * 128 + 255) / 256 * 256, B_TAP = BN * 128, B_BYTES = TPS * B_TAP;
  constexpr int A_IT = (HP + RPP - 1) / RPP, B_IT = BN / RPP;
  constexpr int MAIN_BYTES = A_BYTES + 2 * B_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN_BYTES + (STAT2 ? 2 : 1) * 2 * 4 * BN * 4];
  unsigned char* As = smem;
  unsigned char* Bs = smem + A_BYTES;
  float* red = reinterpret_cast<float*>(smem + MAIN_BYTES);       // [tile parity][consumer wave][BN] column sums (STAT2: followed by the same of the squares)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntiles = p.n_tiles_m * p.n_tiles_n;
  int t, t_end, t_stride;
  if ((gridDim.x & 7) == 0) {
    const int x = blockIdx.x & 7, q = ntiles >> 3, r = ntiles & 7;
    const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    t = start + (blockIdx.x >> 3);
    t_end = start + q + (x < r ? 1 : 0);
    t_stride = gridDim.x >> 3;
  } else {
    t = blockIdx.x;
    t_end = ntiles;
    t_stride = gridDim.x;
  }
  if (t >= t_end) return;
  const int nchunks = p.ld_in / KCH;
  const int tiles_per_img = p.h / R;
  const int half = lane >> 5, col = lane & 31;

  if (wave >= 4) {
    // ================================================= loaders =====================================================================
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
    const int lt = tid - 256;
    const int qu = lt % UPR, row_t = lt / UPR;
    const int st_lds = BF16 ? lds_off(row_t, qu >> 1) + (qu & 1) * 8 : lds_off(row_t, qu);
    uint32_t a_voff[A_IT];
    uint32_t b_voff;
    const uint32_t b_pass = (uint32_t)(RPP * p.w_sn * 4);
    auto set_tile_a = [&](int tt) {
      const int mt = tt / p.n_tiles_n;
      const int img = mt / tiles_per_img, row0 = (mt - img * tiles_per_img) * R;
      int rt = row_t;
      asm volatile("" : "+v"(rt));                              // keeps the per-pass halo coordinates from being hoisted out of the tile loop (22 x 2 registers)
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int hp = rt + RPP * i;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int iy = row0 + hy - 1, ix = hx - 1;
        const bool ok = hp < HP && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)W;
        a_voff[i] = ok ? (uint32_t)(((img * p.h + iy) * W + ix) * p.ld_in + 4 * qu) * 4u : OOB;
      }
    };
    int nt_b = 0, bbuf_w = 0;                                   // filter column tile the fetches read; filter buffer the next fetch fills
    auto set_tile_b = [&](int tt) {
      nt_b = tt % p.n_tiles_n;
      b_voff = (uint32_t)(((int64_t)(nt_b * BN + row_t) * p.w_sn + 4 * qu) * 4);
    };
    // The next halo waits in registers from step g = 1 (loads issued, after that step's filter group has been written, so that the two
    // register images are never live together) to the end of step g = 2; with bf16 operands it is packed at the top of step 2
    // (88 -> 44 registers) before that step's filter loads take their 96.
    u32x4 ra[A_IT], rb[BF16 ? 1 : TPS * B_IT];
    u32x2 rap[BF16 ? A_IT : 1];
    auto gload_a = [&](int c0) {
      const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(c0 * 4);
#pragma unroll
      for (int i = 0; i < A_IT; ++i)
        if (!(TG_ABL & 8) || i < A_IT / 2) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, a_voff[i], so, 0);
    };
    auto pack_a = [&]() {
      if constexpr (BF16) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) rap[i] = pack4(ra[i]);
      }
    };
    auto put = [&](unsigned char* dst, u32x4 v) {
      if constexpr (BF16) *reinterpret_cast<u32x2*>(dst) = pack4(v);
      else *reinterpret_cast<u32x4*>(dst) = v;
    };
    auto sstore_a = [&]() {
#pragma unroll
      for (int i = 0; i < A_IT; ++i)
        if (i + 1 < A_IT || row_t + RPP * i < HP) {
          if constexpr (BF16) *reinterpret_cast<u32x2*>(As + st_lds + i * RPP * 128) = rap[i];
          else *reinterpret_cast<u32x4*>(As + st_lds + i * RPP * 128) = ra[i];
        }
    };
    // filter group (three taps) of the next step.  fp32 operands: global -> registers (gload_b) -> LDS (sstore_b).  bf16 operands: the
    // packed images go straight to LDS (gload_b: 12 LDS-DMA instructions of 1 KB per loader wave), nothing left for sstore_b.
    auto gload_b = [&](int g, int c0) {
      if constexpr (BF16) {
        const unsigned char* src = static_cast<const unsigned char*>(p.wpk) + ((size_t)((nt_b * nchunks + c0 / KCH) * NG + g)) * B_BYTES + (wave - 4) * (B_BYTES / 4) + lane * 16;
        unsigned char* dst = Bs + bbuf_w * B_BYTES + (wave - 4) * (B_BYTES / 4);
#pragma unroll
        for (int j = 0; j < B_BYTES / 4 / 1024; ++j)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 1024), (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                      // barrier_keep<A_IT> counts on the halo loads being issued AFTER these
      } else {
#pragma unroll
        for (int k = 0; k < TPS; ++k) {
          const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(((p.tap[TPS * g + k] >> 16) * (int)p.w_st + c0) * 4);
#pragma unroll
          for (int j = 0; j < B_IT; ++j)
            if (!(TG_ABL & 16) || j < B_IT / 2) rb[k * B_IT + j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, b_voff, so + j * b_pass, 0);
        }
      }
    };
    auto sstore_b = [&](int buf) {
      if constexpr (!BF16) {
        unsigned char* b = Bs + buf * B_BYTES + st_lds;
#pragma unroll
        for (int k = 0; k < TPS; ++k)
#pragma unroll
          for (int j = 0; j < B_IT; ++j) put(b + k * B_TAP + j * RPP * 128, rb[k * B_IT + j]);
      }
    };
    auto flush = [&](int tt, int parity) {                      // column sums of tile tt: the four consumer slots -> one double atomic per channel
      if (lt < BN) {
        const int nt = tt % p.n_tiles_n, m0 = (tt / p.n_tiles_n) * BM;
        int seg = 0, acc_rows = p.seg_rows[0];
        while (seg < p.nseg - 1 && m0 >= acc_rows) acc_rows += p.seg_rows[++seg];
        const float* rp = red + parity * 4 * BN + lt;
        const float s1 = (rp[0] + rp[BN]) + (rp[2 * BN] + rp[3 * BN]);
        if constexpr (STAT2) {
          const float* rq = rp + 2 * 4 * BN;
          const float s2 = (rq[0] + rq[BN]) + (rq[2 * BN] + rq[3 * BN]);
          if (nt * BN + lt < p.n_store) {
            atomicAdd(p.colsum + ((int64_t)seg * 2) * p.c_out + nt * BN + lt, (double)s1);
            atomicAdd(p.colsum + ((int64_t)seg * 2 + 1) * p.c_out + nt * BN + lt, (double)s2);
          }
        } else if (nt * BN + lt < p.n_store) atomicAdd(p.colsum + (int64_t)seg * p.c_out + nt * BN + lt, (double)s1);
      }
    };
    set_tile_a(t);
    set_tile_b(t);
    gload_b(0, 0);
    gload_a(0);
    pack_a();
    sstore_a();
    sstore_b(0);
    barrier_keep<0>();                                          // (P) operands of the first step are in LDS
    int bbuf = 0, k_tile = 0;
    while (true) {
      const int tn = t + t_stride;
      const bool has_next = tn < t_end;
      for (int c = 0; c < nchunks; ++c) {
        const bool last_c = c + 1 == nchunks;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const bool last_g = g == NG - 1;
          const bool more = !(last_g && last_c) || has_next;
          bool halo_in_flight = false;
          if (more) {
            if (last_g) pack_a();
            bbuf_w = bbuf ^ 1;                                  // the other filter buffer: last read one step ago
            if (!last_g) {
              gload_b(g + 1, c * KCH);
            } else if (!last_c) {
              gload_b(0, (c + 1) * KCH);
            } else {
              set_tile_b(tn);
              gload_b(0, 0);
            }
            sstore_b(bbuf ^ 1);
            if (g == 1) {
              if (!last_c) {
                gload_a((c + 1) * KCH);
                halo_in_flight = true;
              } else if (has_next) {
                set_tile_a(tn);
                gload_a(0);
                halo_in_flight = true;
              }
            }
          }
          // (S) the consumers are done with this step; the filter fetch has landed, the halo loads (issued after it) stay in flight
          if (halo_in_flight) barrier_keep<A_IT>();
          else barrier_keep<0>();
          if (COLSUM && c == 0 && g == 0 && k_tile > 0) flush(t - t_stride, (k_tile - 1) & 1);
          if (last_g && more) {
            sstore_a();                                         // the ONE halo buffer: free now
            barrier_keep<0>();                                  // (H)
          }
          bbuf ^= 1;
        }
      }
      if (!has_next) break;
      t = tn;
      ++k_tile;
    }
    if (COLSUM) {
      barrier_keep<0>();                                        // (F) the consumers' last column sums are in LDS
      flush(t, k_tile & 1);
    }
  } else {
    // ================================================= consumers ===================================================================
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ymul ? p.ymul : p.out), 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias ? p.bias : p.w), 0, p.bias ? (uint32_t)p.n_store * 4u : 0u, 0x00020000);
    const bool ym = COLSUM && !STAT2 && p.ymul != nullptr;
    const int wm0 = wave * 64;
    // Which pixel of its 32-pixel fragment lane `col` multiplies.  A ds_read_b128 is served in four groups of 16 lanes ({0-3, 12-15, 20-27},
    // {4-11, 16-19, 28-31} and the same + 32), conflict-free when the 16 halo rows of a group differ mod 16 (128-byte rows, the swizzle
    // key is row bits 1..3).  W = 32 / 64: the 32 pixels are consecutive in one halo row -> they do.  W = 16: lanes 16-31 sit in the
    // NEXT image row, 18 halo rows further on, and raster order makes rows 12, 13 mod 16 collide with the first row's (measured: 23 % of
    // the LDS cycles were conflicts); rotating the second row's pixels by two (lane 16 + i takes x = (i - 2) mod 16) restores
    // row = base + i mod 16.  The epilogue stores through the same map (frag_pixel).
    const int pc = (W == 16 && col >= 16) ? 16 + ((col - 18) & 15) : col;
    int a_hp[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int r = wm0 + mi * 32 + pc;
      const int ty = r / W, tx = r - ty * W;
      a_hp[mi] = (ty + 1) * HW_ + tx + 1;
    }
    int b_off[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) b_off[ni] = (ni * 32 + col) * 128;
    const int b_swz = (col >> 1) & 7;
    f32x16 acc[2][4];
#ifdef TG_STAMP
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, sum_k = 0, sum_e = 0, sum_b = 0, tb0 = 0, tb1 = 0;
    LSTAMP(ts0);
#endif
    __syncthreads();                                            // (P)
#ifdef TG_STAMP
    LSTAMP(ts1);
#endif
    int bbuf = 0, k_tile = 0;
    while (true) {
      const int tn = t + t_stride;
      const bool has_next = tn < t_end;
#ifdef TG_STAMP
      LSTAMP(ts2);
#endif
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
      for (int c = 0; c < nchunks; ++c) {
        const bool last_c = c + 1 == nchunks;
#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
          const bool last_g = g == NG - 1;
          const bool more = !(last_g && last_c) || has_next;
          const unsigned char* B = Bs + bbuf * B_BYTES;
          if constexpr (TG_C3_PIPE == 2 || (TG_C3_PIPE == 1 && BF16)) {
            // A wave issues in order: six fragment reads in front of the MFMAs of a (tap, k-slice) sub-step left the pipe idle for an LDS
            // latency, ~90 cycles per sub-step (stamps, conv1_2: fp32 26 084 cycles per step against 24 576 of MFMAs, bf16 4 829 against
            // 3 072 with 774 at the barriers).  Here the twelve sub-steps of a step are one software pipeline over two fragment register
            // sets: the six reads of sub-step i + 1 are issued one each behind the first six MFMAs of sub-step i (in the order their
            // MFMAs need them), so only the first reads of a step — behind its barrier — are waited for.  bf16: +1 ... 5 % (the launch is paced
            // by its loaders); fp32 (TG_C3_PIPE=2): 1.5 - 2 % slower than the compiler's order, not shipped.
            using frag_t = typename std::conditional<BF16, bf16x8, f32x4>::type;
            constexpr int NM = BF16 ? 8 : 32;                     // MFMAs per sub-step: 2 x 4 tiles (x 4 k-pairs of the fp32 fragment)
            int a_row[TPS][2], a_swz[TPS][2];
#pragma unroll
            for (int k = 0; k < TPS; ++k) {
              const int tp = p.tap[TPS * g + k];
              const int shift = (int)(int8_t)(tp >> 8) * HW_ + (int)(int8_t)tp;
#pragma unroll
              for (int mi = 0; mi < 2; ++mi) {
                const int hp = a_hp[mi] + shift;
                a_row[k][mi] = hp * 128;
                a_swz[k][mi] = (hp >> 1) & 7;
              }
            }
            frag_t fa[2][2], fb[2][4];
            auto rd = [&](int i, int j, int set) {                // read j of sub-step i: a0, b0, b1, b2, b3, a1
              const int k = i >> 2, s = i & 3;
              if (j == 0 || j == 5) {
                const int mi = j == 0 ? 0 : 1;
                fa[set][mi] = *reinterpret_cast<const frag_t*>(As + a_row[k][mi] + (((2 * s + half) ^ a_swz[k][mi]) << 4));
              } else {
                fb[set][j - 1] = *reinterpret_cast<const frag_t*>(B + k * B_TAP + b_off[j - 1] + (((2 * s + half) ^ b_swz) << 4));
              }
            };
#pragma unroll
            for (int j = 0; j < 6; ++j) rd(0, j, 0);
#pragma unroll
            for (int i = 0; i < 4 * TPS; ++i) {
#pragma unroll
              for (int q = 0; q < NM; ++q) {
                const int mi = (q >> 2) & 1, ni = q & 3;
                if constexpr (BF16) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i & 1][mi], fb[i & 1][ni], acc[mi][ni], 0, 0, 0);   // D[pixel][channel]: lane = channel
                else acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i & 1][mi][q >> 3], fb[i & 1][ni][q >> 3], acc[mi][ni], 0, 0, 0);
                if (i + 1 < 4 * TPS && q < 6) {
                  __builtin_amdgcn_sched_barrier(0);
                  rd(i + 1, q, (i + 1) & 1);
                  __builtin_amdgcn_sched_barrier(0);
                }
              }
            }
          } else
#pragma unroll
          for (int k = 0; k < TPS; ++k) {
            const int tp = p.tap[TPS * g + k];
            const int shift = (int)(int8_t)(tp >> 8) * HW_ + (int)(int8_t)tp;
            int a_row[2], a_swz[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
              const int hp = a_hp[mi] + shift;
              a_row[mi] = hp * 128;
              a_swz[mi] = (hp >> 1) & 7;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              if constexpr (BF16) {
                bf16x8 a[2], b[4];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(As + a_row[mi] + (((2 * s + half) ^ a_swz[mi]) << 4));
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(B + k * B_TAP + b_off[ni] + (((2 * s + half) ^ b_swz) << 4));
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                  for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);   // D[pixel][channel]: lane = channel
              } else {
                f32x4 a[2], b[4];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(As + a_row[mi] + (((2 * s + half) ^ a_swz[mi]) << 4));
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(B + k * B_TAP + b_off[ni] + (((2 * s + half) ^ b_swz) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][e], b[ni][e], acc[mi][ni], 0, 0, 0);
              }
            }
          }
#ifdef TG_STAMP
          LSTAMP(tb0);
#endif
          __syncthreads();                                      // (S)
          if (last_g && more) __syncthreads();                  // (H)
#ifdef TG_STAMP
          LSTAMP(tb1);
          sum_b += tb1 - tb0;
#endif
          bbuf ^= 1;
        }
      }
#ifdef TG_STAMP
      LSTAMP(ts3);
      sum_k += ts3 - ts2;
#endif
      // ---- epilogue of tile t: registers -> global.  Tiles are whole rows of whole images, so the tile's pixels are consecutive rows
      // of the [pixels][ld_out] output.  D[pixel][channel]: lane = channel ni*32 + col, register r of fragment mi = pixel
      // mi*32 + (r & 3) + 8*(r >> 2) + 4*half, so one dword store writes two whole 128-byte lines (one per half-wave).  Measured
      // (tools/micro/store_patterns.hip, one workgroup per CU storing such tiles): this shape and 16-byte stores of whole lines both
      // reach ~48 B/clk/CU when few CUs store and the chip-wide ~6 TB/s when all do; 16-byte stores straight from the transposed
      // accumulator (32 B per line and instruction) stay at 15 B/clk/CU.  Bias, the activation and the column sums are per lane.
      {
        const int n0 = (t % p.n_tiles_n) * BN;
        uint32_t px0 = (uint32_t)((t / p.n_tiles_n) * BM + wm0 + 4 * half);
        asm volatile("" : "+v"(px0));                             // per-tile value: keeps the 32 row offsets below out of the registers during the K loop
        const uint32_t row_bytes = (uint32_t)p.ld_out * 4u;
        // address = per-lane offset (this lane's channel in the wave's first pixel row; out of range beyond n_store) + a wave-uniform
        // scalar offset per (mi, r): four address registers instead of thirty-two
        uint32_t ch_off[4];
        float cs[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int n = n0 + ni * 32 + col;
          ch_off[ni] = n < p.n_store ? px0 * row_bytes + (uint32_t)n * 4u : OOB;
          cs[ni] = 0.f;
        }
        // pixel of fragment row (r & 3) + 8*(r >> 2) + 4*half, less the 4*half that px0 carries; W = 16: rows 16.. are rotated (see pc
        // above) and for r = 8, 9 the lower half-wave's pixels wrap around to the end of the image row (+16: wrap16)
        auto pix_off = [&](int mi, int r) {
          const int i = (W == 16 && r >= 8) ? 16 + 8 * ((r >> 2) - 2) + (r & 3) - 2 : (r & 3) + 8 * (r >> 2);
          return (uint32_t)(mi * 32 + i) * row_bytes;
        };
        const uint32_t wrap16 = (W == 16 && half == 0) ? 16u * row_bytes : 0u;
        auto lane_off = [&](int ni, int r) { return (W == 16 && (r == 8 || r == 9)) ? ch_off[ni] + wrap16 : ch_off[ni]; };
        if (!COLSUM) {
          // bias + activation; only y = x > 0 ? x : slope * x forms reach this kernel (none / relu / leaky relu: conv3x3_bf16_launch)
          const float slope = p.act == TG_ACT_LRELU ? p.alpha : (p.act == TG_ACT_RELU ? 0.f : 1.f);
          float bias_v[4];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const uint32_t bb = __builtin_amdgcn_raw_buffer_load_b32(rs_b, (uint32_t)(n0 + ni * 32 + col) * 4u, 0, 0);      // beyond n_store / no bias: 0
            bias_v[ni] = __builtin_bit_cast(float, bb);
          }
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const uint32_t off = pix_off(mi, r);
#pragma unroll
              for (int ni = 0; ni < 4; ++ni) {
                const float x = acc[mi][ni][r] + bias_v[ni];
                const float v = x > 0.f ? x : slope * x;
                if (!(TG_ABL & 1)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs_o, lane_off(ni, r), off, 0);
              }
            }
        } else if (STAT2) {
          float cq[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.stat2 == 2) {
            // input gradient that a batch norm's backward pass consumes (tg_igemm_bnbwdstat_*): the stored value is the gradient dy itself and
            // the statistics are that pass's two sums, of dy and of dy * x, x = the batch norm's input read at the output's own addresses
            // (one (mi, ni) fragment ahead of its use, as the multiplier below)
            uint32_t xv[2][16];
            auto xload = [&](int f, int buf) {
              const int mi = f >> 2, ni = f & 3;
#pragma unroll
              for (int r = 0; r < 16; ++r) xv[buf][r] = __builtin_amdgcn_raw_buffer_load_b32(rs_y, lane_off(ni, r), pix_off(mi, r), 0);
            };
            xload(0, 0);
#pragma unroll
            for (int f = 0; f < 8; ++f) {
              const int mi = f >> 2, ni = f & 3;
              if (f + 1 < 8) xload(f + 1, (f + 1) & 1);
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const float v = acc[mi][ni][r];
                cs[ni] += v;
                cq[ni] += v * __builtin_bit_cast(float, xv[f & 1][r]);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs_o, lane_off(ni, r), pix_off(mi, r), 0);
              }
            }
          } else {
            // batch norm behind the layer: the stored value is v = act(acc + bias) and the statistics are of v (sum and sum of squares per
            // application segment and channel: the tf.nn.moments / fused batch-norm pass over the activation disappears)
            const float slope = p.act == TG_ACT_LRELU ? p.alpha : (p.act == TG_ACT_RELU ? 0.f : 1.f);
            float bias_v[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
              const uint32_t bb = __builtin_amdgcn_raw_buffer_load_b32(rs_b, (uint32_t)(n0 + ni * 32 + col) * 4u, 0, 0);
              bias_v[ni] = __builtin_bit_cast(float, bb);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const uint32_t off = pix_off(mi, r);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                  const float x = acc[mi][ni][r] + bias_v[ni];
                  const float v = x > 0.f ? x : slope * x;
                  cs[ni] += v;
                  cq[ni] += v * v;
                  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs_o, lane_off(ni, r), off, 0);
                }
              }
          }
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const float o = __shfl_xor(cq[ni], 32);
            if (half == 0) red[2 * 4 * BN + ((k_tile & 1) * 4 + wave) * BN + ni * 32 + col] = cq[ni] + o;
          }
        } else if (ym) {
          // the multiplier act'(y) is read at the output's own addresses, one (mi, ni) fragment (16 dwords per lane) ahead of its use
          const float slope = p.ymul_act == TG_ACT_LRELU ? p.ymul_alpha : (p.ymul_act == TG_ACT_RELU ? 0.f : 1.f);
          uint32_t yv[2][16];
          auto yload = [&](int f, int buf) {
            const int mi = f >> 2, ni = f & 3;
#pragma unroll
            for (int r = 0; r < 16; ++r) yv[buf][r] = __builtin_amdgcn_raw_buffer_load_b32(rs_y, lane_off(ni, r), pix_off(mi, r), 0);
          };
          yload(0, 0);
#pragma unroll
          for (int f = 0; f < 8; ++f) {
            const int mi = f >> 2, ni = f & 3;
            if (f + 1 < 8) yload(f + 1, (f + 1) & 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const uint32_t yb = yv[f & 1][r];
              const float v = acc[mi][ni][r] * (__builtin_bit_cast(float, yb) > 0.f ? 1.f : slope);
              cs[ni] += v;
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs_o, lane_off(ni, r), pix_off(mi, r), 0);
            }
          }
        } else {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const uint32_t off = pix_off(mi, r);
#pragma unroll
              for (int ni = 0; ni < 4; ++ni) {
                const float v = acc[mi][ni][r];
                cs[ni] += v;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs_o, lane_off(ni, r), off, 0);
              }
            }
        }
        if (COLSUM) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const float o = __shfl_xor(cs[ni], 32);
            if (half == 0) red[((k_tile & 1) * 4 + wave) * BN + ni * 32 + col] = cs[ni] + o;
          }
        }
      }
#ifdef TG_STAMP
      LSTAMP(ts2);
      sum_e += ts2 - ts3;
#endif
      if (!has_next) break;
      t = tn;
      ++k_tile;
    }
#ifdef TG_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 64) {                  // consumer wave 0: first barrier wait | K loops | of which at barriers | epilogues | total | tiles
      tg_conv_stamps[0 * 64 + blockIdx.x] = ts1 - ts0;
      tg_conv_stamps[1 * 64 + blockIdx.x] = sum_k;
      tg_conv_stamps[2 * 64 + blockIdx.x] = sum_b;
      tg_conv_stamps[3 * 64 + blockIdx.x] = sum_e;
      tg_conv_stamps[4 * 64 + blockIdx.x] = ts2 - ts0;
      tg_conv_stamps[5 * 64 + blockIdx.x] = (unsigned long long)(k_tile + 1);
    }
#endif
    if (COLSUM) __syncthreads();                                // (F)
  }
}

template <int W>
void launch_pipe(ConvParams& p, hipStream_t s) {
  const int tiles = p.n_tiles_m * p.n_tiles_n, cus = compute_units();
  const dim3 grid(tiles < cus ? tiles : cus);                  // one resident workgroup per CU (LDS), each walking its share of the tiles
  if (p.f32) {
    if (p.colsum && p.stat2) hipLaunchKernelGGL((conv3x3_pipe_kernel<W, true, false, true>), grid, dim3(512), 0, s, p);
    else if (p.colsum) hipLaunchKernelGGL((conv3x3_pipe_kernel<W, true, false>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((conv3x3_pipe_kernel<W, false, false>), grid, dim3(512), 0, s, p);
  } else {
    if (p.colsum && p.stat2) hipLaunchKernelGGL((conv3x3_pipe_kernel<W, true, true, true>), grid, dim3(512), 0, s, p);
    else if (p.colsum) hipLaunchKernelGGL((conv3x3_pipe_kernel<W, true, true>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((conv3x3_pipe_kernel<W, false, true>), grid, dim3(512), 0, s, p);
  }
}

// A/B switches, read once at library load: the generic implicit GEMM takes the launches instead
const bool g_disabled = getenv("TG_NO_CONV3X3_BF16") != nullptr;
const bool g_disabled_f32 = getenv("TG_NO_CONV3X3_F32") != nullptr;

// bytes of the packed bf16 filter of a launch (filter_pack_kernel): (c_out / 128) x (ld_in / 64) x 9 images of 16 KB
int64_t pack_bytes(const tg_igemm_desc* d) { return (int64_t)(d->c_out / BN) * (d->ld_in / KC) * 9 * (BN * 128); }

int g_policy = 0;                      // tg_conv3x3_policy: 0 = where it pays (below), 1 = wherever it applies, 2 = never
long g_launches = 0;

// the one tile shape: 256 output pixels = whole image rows
bool rows_fit(const tg_igemm_desc* d) { return d->w_in > 0 && 256 % d->w_in == 0 && d->h_in % (256 / d->w_in) == 0; }

}  // namespace

namespace tg {

int halo_policy() { return g_policy; }
int halo_compute_units() { return compute_units(); }
void halo_count_launch() { ++g_launches; }

// the layer has the kernel's shape (independent of how many images the launch holds)
static bool conv3x3_fits(const tg_igemm_desc* d, int n_desc, const int32_t* seg_rows, int nseg, bool bf16) {
  if ((bf16 ? g_disabled : g_disabled_f32) || g_policy == 2) return false;
  if (n_desc != 1 || d->n_taps != 9 || d->n_group != 0) return false;
  if (d->s_y != 1 || d->s_x != 1 || d->os_y != 1 || d->os_x != 1 || d->oo_y != 0 || d->oo_x != 0) return false;
  if (d->h_v != d->h_in || d->w_v != d->w_in || d->h_out != d->h_in || d->w_out != d->w_in) return false;
  if (d->w_in != 16 && d->w_in != 32 && d->w_in != 64) return false;
  if (!rows_fit(d)) return false;
  if (d->ld_in % (bf16 ? KC : 32) || d->c_out % BN) return false;
  const auto simple = [](int a) { return a == TG_ACT_NONE || a == TG_ACT_LRELU || a == TG_ACT_RELU; };
  if (!simple(d->act)) return false;                          // the register epilogue applies none / relu / leaky relu
  bool seen[9] = {false, false, false, false, false, false, false, false, false};
  for (int t = 0; t < 9; ++t) {                               // the nine taps of a 3x3 window, each exactly once, in any order
    if (d->dy[t] < -1 || d->dy[t] > 1 || d->dx[t] < -1 || d->dx[t] > 1) return false;
    const int k = (d->dy[t] + 1) * 3 + d->dx[t] + 1;
    if (seen[k]) return false;
    seen[k] = true;
  }
  const int per_img = d->h_in * d->w_in;
  for (int i = 0; i < nseg; ++i)
    if (seg_rows[i] % per_img) return false;                  // applications are whole images: a tile never straddles a segment
  return true;
}

// One workgroup per CU (141-150 KB of LDS), so a launch runs in ceil(tiles / CUs) rounds and the last one may be nearly empty: 130 images
// of 32x32 are 520 tiles = 3 rounds on 256 CUs for 2.03 rounds of work.  The generic implicit GEMM (thousands of small workgroups) has no
// such step, so the halo form is taken only where its per-tile advantage (measured on full rounds: 1.07x with fp32 operands, 1.8x with
// bf16 ones) survives the quantisation.
static long conv3x3_slots(const tg_igemm_desc* d, bool bf16) { (void)d; (void)bf16; return (long)compute_units(); }
static long conv3x3_tiles_per_image(const tg_igemm_desc* d, bool bf16) { (void)bf16; return (long)d->h_in * d->w_in / 256 * (d->c_out / BN); }

static bool conv3x3_pays(const tg_igemm_desc* d, bool bf16, int n_img) {
  if (g_policy != 0) return true;
  const long slots = conv3x3_slots(d, bf16), tiles = conv3x3_tiles_per_image(d, bf16) * n_img;
  const long rounds = (tiles + slots - 1) / slots;
  return (double)tiles / (double)(rounds * slots) * (bf16 ? 1.8 : 1.07) >= 1.0;
}

bool conv3x3_bf16_applicable(const tg_igemm_desc* d, int n_desc, const int32_t* seg_rows, int nseg, bool bf16) {
  return conv3x3_fits(d, n_desc, seg_rows, nseg, bf16) && conv3x3_pays(d, bf16, d->n_img);
}

// A launch of the right shape whose last round is less than 90 % full: the number of LEADING images that make whole rounds — they go to
// the halo kernel, the few images left to the generic one (igemm_impl splits the launch; 130 images of 32x32: 128 + 2).  0: no such
// split (rounds full enough, too few images for one round, or the whole-round prefix is less than 60 % of the launch).
int conv3x3_bf16_split_images(const tg_igemm_desc* d, int n_desc, const int32_t* seg_rows, int nseg, bool bf16) {
  if (g_policy != 0 || !conv3x3_fits(d, n_desc, seg_rows, nseg, bf16)) return 0;
  const long slots = conv3x3_slots(d, bf16), tpi = conv3x3_tiles_per_image(d, bf16);
  const long tiles = tpi * d->n_img, rounds = (tiles + slots - 1) / slots;
  if (tiles * 10 >= rounds * slots * 9) return 0;
  const long full = tiles / slots;                            // whole rounds in the launch
  if (full < 1) return 0;
  const long head = full * slots / tpi;                       // images in them (rounded down: the head's last round may miss a few tiles)
  if (head < 1 || head >= d->n_img || head * 10 < (long)d->n_img * 6 || !conv3x3_pays(d, bf16, (int)head)) return 0;
  return (int)head;
}

int64_t conv3x3_bf16_pack_bytes(const tg_igemm_desc* d, int n_desc) {
  const int32_t whole = d->n_img * d->h_in * d->w_in;         // shape only: one segment of whole images
  if (g_disabled || !conv3x3_fits(d, n_desc, &whole, 0, true)) return 0;
  return pack_bytes(d);
}

int conv3x3_bf16_launch(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, float* out, double* colsum,
                        const int32_t* seg_rows, int nseg, const float* ymul, int ymul_act, float ymul_alpha, uint32_t in_bytes, uint32_t w_bytes,
                        uint32_t out_bytes, hipStream_t s, bool bf16, void* scratch, int64_t scratch_bytes, int stat2) {
  ConvParams p;
  p.stat2 = stat2;
  p.in = in; p.w = w; p.bias = bias; p.out = out; p.colsum = colsum; p.ymul = ymul; p.ymul_act = ymul_act; p.ymul_alpha = ymul_alpha;
  p.nseg = nseg;
  for (int i = 0; i < 8; ++i) p.seg_rows[i] = (seg_rows && i < nseg) ? seg_rows[i] : 0;
  p.n_img = d->n_img; p.h = d->h_in; p.ld_in = d->ld_in; p.ld_out = d->ld_out; p.c_out = d->c_out; p.n_store = d->n_store;
  p.act = d->act; p.alpha = d->alpha;
  p.w_sn = d->w_sn; p.w_st = d->w_st;
  for (int t = 0; t < 9; ++t) p.tap[t] = ((int)d->tapw[t] << 16) | (((int)d->dy[t] & 0xff) << 8) | ((int)d->dx[t] & 0xff);
  p.f32 = bf16 ? 0 : 1;
  p.n_tiles_m = d->n_img * d->h_in * d->w_in / 256;
  p.n_tiles_n = d->c_out / BN;
  p.in_bytes = in_bytes; p.w_bytes = w_bytes; p.out_bytes = out_bytes;
  const auto simple = [](int a) { return a == TG_ACT_NONE || a == TG_ACT_LRELU || a == TG_ACT_RELU; };
  TG_REQUIRE(ymul == nullptr || simple(ymul_act), "conv3x3: activation %d of the gradient multiplier is not none / relu / leaky relu", ymul_act);
  p.wpk = nullptr;
  if (bf16) {
    // the bf16 filter goes global -> LDS by LDS-DMA from a packed copy in CALLER-OWNED scratch (size: tg_igemm_workspace_bytes(descs, n_desc, seg_rows, nseg, bf16 = 1));
    // the library allocates nothing, and a launch without the scratch it was told to bring is an error, not a slower kernel
    const int64_t need = pack_bytes(d);
    TG_REQUIRE(scratch != nullptr && scratch_bytes >= need,
               "conv3x3 (bf16): this launch needs %lld bytes of scratch for the packed filter, got %lld (query tg_igemm_workspace_bytes(..., bf16 = 1))",
               (long long)need, (long long)(scratch ? scratch_bytes : 0));
    TG_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 15) == 0, "conv3x3 (bf16): scratch must be 16-byte aligned");
    p.wpk = scratch;
    const int nchunks = d->ld_in / KC;
    const int total = p.n_tiles_n * nchunks * 9 * 1024;        // 16-byte chunks of the packed filter
    hipLaunchKernelGGL(filter_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, p, nchunks, total);
    TG_CHECK_LAUNCH("filter_pack_kernel");
  }
  ++g_launches;
  // measured (N = 250 images, conv1_2 / conv2_1 / conv2_2): bf16 869 / 739 / 961 TFLOP/s, with column sums 794 / 672 / 883, exact fp32 123 / 132 / 141
  if (d->w_in == 16) launch_pipe<16>(p, s);
  else if (d->w_in == 32) launch_pipe<32>(p, s);
  else launch_pipe<64>(p, s);
  TG_CHECK_LAUNCH("conv3x3_pipe_kernel");
  return TG_OK;
}

}  // namespace tg

extern "C" int tg_conv3x3_policy(int policy) {
  const int was = g_policy;
  if (policy >= 0 && policy <= 2) g_policy = policy;
  return was;
}

extern "C" int64_t tg_conv3x3_launches(void) { return g_launches; }

#ifdef TG_STAMP
extern "C" int tg_debug_read_conv_stamps(unsigned long long* out) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_conv_stamps), sizeof(unsigned long long) * 8 * 64);
  return e == hipSuccess ? 0 : -2;
}
#endif
