// pixel_ops.h - per-element device functions shared by every conv translation unit (conv.hip, conv_bf16.hip, wino.hip):
// the leaky-ReLU of the BN+FiLM prologue / epilogue and the complex-ratio-mask output head.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace {

__device__ __forceinline__ float leaky(float v) { return fmaxf(v, 0.01f * v); }  // == v > 0 ? v : 0.01 v

// Value of lane ^ 1 (the horizontal neighbour in the 2x2 pooling windows): a DPP quad permutation [1,0,3,2] on the VALU instead
// of __shfl_xor's ds_bpermute_b32 through the LDS pipe, which the MFMA operand reads of the co-resident workgroups saturate.
__device__ __forceinline__ float lane_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}


// The complex ratio mask of one time-frequency bin from its three after_conv logits (resunet.py:476-507; torchlibrosa
// magphase clamps |M| at 1e-10); bin 512 is the zero padding of resunet.py:573, whose output is exactly 0.
__device__ __forceinline__ void mask_pixel(const ConvArgs& p, int b, int t, int f, float l0, float l1, float l2) {
    const size_t row = ((size_t)b * p.mask_T + t) * p.mask_nbins + f;
    const float mask_mag = 1.f / (1.f + expf(-l0));
    const float mr = tanhf(l1), mi = tanhf(l2);
    const float mm = sqrtf(mr * mr + mi * mi);
    const float den = fmaxf(mm, 1e-10f);
    const float mc = mr / den, ms = mi / den;
    const float ci = p.mask_cos[row], si = p.mask_sin[row];
    const float oc = ci * mc - si * ms;
    const float os = si * mc + ci * ms;
    const float om = fmaxf(p.mask_mag[row] * mask_mag, 0.f);
    p.mask_re[row] = om * oc;
    p.mask_im[row] = om * os;
    if (f == p.mask_nbins - 2) {  // last kept bin: also write the dropped Nyquist bin's exact zero
        p.mask_re[row + 1] = 0.f;
        p.mask_im[row + 1] = 0.f;
    }
}

}  // namespace
