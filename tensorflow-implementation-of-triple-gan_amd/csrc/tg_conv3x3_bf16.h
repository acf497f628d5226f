// Internal: the halo-tiled bf16 3x3 convolution (conv3x3_bf16.hip) that igemm_impl (igemm.hip) routes matching bf16 launches to.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tg_kernels.h"

namespace tg {

// 9 taps of a 3x3 window, stride 1, output grid = input grid (SAME), width 16 / 32 / 64 with 256 / width dividing the height, 64 | ld_in,
// 128 | c_out, segments of whole images
bool conv3x3_bf16_applicable(const tg_igemm_desc* d, int n_desc, const int32_t* seg_rows, int nseg, bool bf16);   // bf16 = false: the exact-fp32 form (32 | ld_in)
// > 0: the launch has the kernel's shape but does not fill whole rounds of one workgroup per CU; that many leading images do
int conv3x3_bf16_split_images(const tg_igemm_desc* d, int n_desc, const int32_t* seg_rows, int nseg, bool bf16);
int conv3x3_bf16_launch(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, float* out, double* colsum,
                        const int32_t* seg_rows, int nseg, const float* ymul, int ymul_act, float ymul_alpha, uint32_t in_bytes, uint32_t w_bytes,
                        uint32_t out_bytes, hipStream_t s, bool bf16, void* scratch, int64_t scratch_bytes, int stat2 = 0);
// bytes of caller-owned scratch a bf16 launch of these descriptors needs for the packed filter (0: the layer never takes the halo kernel)
int64_t conv3x3_bf16_pack_bytes(const tg_igemm_desc* d, int n_desc);

// shared by the halo kernels (conv3x3_bf16.hip owns the state): tg_conv3x3_policy's value, the device's CU count, the launch counter
int halo_policy();
int halo_compute_units();
void halo_count_launch();

// wgrad3x3.hip: the filter gradient of the same layers with the activation tile read once for the nine taps
bool wgrad3x3_applicable(const tg_igemm_desc* d, int n_split, bool bf16, int policy, int compute_units);
int wgrad3x3_splits(const tg_igemm_desc* d, bool bf16, int policy, int compute_units);
int wgrad3x3_launch(const tg_igemm_desc* d, const float* in, const float* dout, float* slab, int n_split, uint32_t in_bytes, uint32_t dout_bytes,
                    hipStream_t s, bool bf16);

}  // namespace tg
