// mfma_operand_hazard.hip - what hipcc's own hazard recognizer does between a vector instruction that writes an MFMA source
// operand and the MFMA (gfx950): compile to ISA and read it -
//     hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only tools/mfma_operand_hazard.hip -o - | grep -B4 v_mfma
// k_ab: v_mul (A) / v_add (B), s_waitcnt, s_nop 0, v_mfma   - two wait states behind the writer of an A / B operand;
// k_c:  v_pk_mul (SrcC), s_waitcnt, s_nop 0, v_mfma         - the same behind a writer of the accumulator input.
// An `asm volatile` MFMA gets no such padding (cdna_hip_programming.md 5.7 item 2): wino32.hip's statements therefore open with
// `s_nop 1`, and tools/audit_wino32_isa.py checks the two wait states in the shipped ISA (tests/test_host_cpu.py).
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_ab(float* o, float x, float y) {
    float a = o[threadIdx.x];
    f32x4 acc = {o[threadIdx.x + 64], o[threadIdx.x + 128], o[threadIdx.x + 192], o[threadIdx.x + 256]};
    asm volatile("s_nop 7");
    float a2 = a * x;   // VALU write of the A operand right in front of the MFMA
    float b2 = a + y;   // ... and of the B operand
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc, 0, 0, 0);
    o[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ void k_c(float* o, float x) {
    float a = o[threadIdx.x], b = o[threadIdx.x + 32];
    f32x4 acc = {o[threadIdx.x + 64], o[threadIdx.x + 128], o[threadIdx.x + 192], o[threadIdx.x + 256]};
    asm volatile("s_nop 7");
    acc[0] *= x; acc[1] *= x; acc[2] *= x; acc[3] *= x;  // VALU write of SrcC right in front
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    o[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
