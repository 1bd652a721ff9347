// wino_common.h - LDS-DMA / barrier / MFMA-step helpers of the Winograd F(2x2,3x3) kernel (wino.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ void lds_barrier() {  // workgroup barrier that does NOT drain vmcnt (LDS-DMA stays in flight)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {  // all but the N youngest vector-memory operations are complete
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// the same with a wave-uniform run-time count (0 ... 15; anything larger waits for 15, which is merely stricter)
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
    switch (n) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<1>(); break;
        case 2: wait_vmcnt<2>(); break;
        case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 5: wait_vmcnt<5>(); break;
        case 6: wait_vmcnt<6>(); break;
        case 7: wait_vmcnt<7>(); break;
        case 8: wait_vmcnt<8>(); break;
        case 9: wait_vmcnt<9>(); break;
        case 10: wait_vmcnt<10>(); break;
        case 11: wait_vmcnt<11>(); break;
        case 12: wait_vmcnt<12>(); break;
        case 13: wait_vmcnt<13>(); break;
        case 14: wait_vmcnt<14>(); break;
        default: wait_vmcnt<15>(); break;
    }
}

typedef int v4i32 __attribute__((ext_vector_type(4)));

// Workgroup -> (spatial tile bx, cout block by, clip bz) from a 1-D grid.  Workgroups are dealt round-robin over the 8
// XCDs (each with its own L2), so with the natural order the gy cout blocks that re-read one input tile land on gy
// different XCDs and the tile is fetched gy times from HBM.  With xcd_map the gy blocks of a (tile, clip) pair are given
// linear ids that are equal mod 8: same XCD, the re-reads hit its L2.  (Speed only: nothing depends on the placement.)
__device__ __forceinline__ void block_coords(const ConvArgs& p, int& bx, int& by, int& bz) {
    const unsigned lin = blockIdx.x;
    unsigned xz, y;
    if (p.xcd_map) {
        const unsigned q = lin & 7u, s = lin >> 3;
        const unsigned g = s / (unsigned)p.gy;
        y = s - g * (unsigned)p.gy;
        // XCD q walks ONE contiguous range of tiles (neighbouring tiles share halo rows and 128-byte lines: under one L2
        // they are fetched once; dealt out tile by tile, xz = g * 8 + q, every XCD fetched them for itself)
        xz = p.xcd_map == 2 ? q * (((unsigned)p.gx * (unsigned)p.B) >> 3) + g : g * 8u + q;
    } else {
        xz = lin / (unsigned)p.gy;
        y = lin - xz * (unsigned)p.gy;
    }
    bz = (int)(xz / (unsigned)p.gx);
    bx = (int)(xz - (unsigned)bz * (unsigned)p.gx);
    by = (int)y;
}

// Buffer descriptor words for inline-asm buffer ops (raw buffer, stride 0, `bytes` records), from wave-uniform inputs.
__device__ __forceinline__ v4i32 make_rsrc_words(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    v4i32 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    r.y = __builtin_amdgcn_readfirstlane((int)(a >> 32));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

// One 1-KiB LDS-DMA piece: 64 lanes x 16 B from (buffer base + scalar byte offset soff + per-lane byte offset) to LDS
// byte address `lds_addr` (wave-uniform, + lane*16).  Scalar offset + constant 32-bit lane offset: no VALU address
// arithmetic per piece (the f32 MFMA shares the SIMD's VALU issue).  Issued from inline asm so that hipcc does not know
// about the pending LDS write: with the builtin it drains vmcnt(0) in front of the next ds_read.  Completion is tracked by
// hand (wait_vmcnt).
__device__ __forceinline__ void lds_dma_16B(v4i32 rsrc, unsigned lane_off, unsigned soff, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_off), "s"(rsrc), "s"(soff), "s"(lds_addr)
        : "memory");
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// One chunk = NS steps of 4 MFMAs (v_mfma_f32_16x16x4_f32): a step covers 2 k-steps (8 input channels) x 2 cout tiles of
// one xi.  Per step a lane reads ONE 16-byte A fragment {(k0,t0), (k0,t1), (k1,t0), (k1,t1)} (ds_read_b128) and ONE
// 8-byte B fragment {k0, k1} (ds_read_b64): 2 LDS instructions and 6 LDS-array cycles per 4 MFMAs, where scalar reads
// took 6 instructions and 12 cycles (ds_read_b32 runs at half the LDS rate).  The reads of step s+PFD are issued before
// the MFMAs of step s (pinned: hipcc sinks them otherwise and then waits lgkmcnt(0) in front of every MFMA).
// The two accumulators of a step alternate, so no MFMA depends on its predecessor (40-cycle dependent latency).
template <int NS, int AST, int BST, typename XiOf>
__device__ __forceinline__ void gemm_steps(const float* afrag, const float* bfrag, f32x4 (&acc)[16][2], XiOf xi_of) {
    constexpr int PFD = 2;
    f32x4 av[PFD + 1];
    f32x2 bv[PFD + 1];
    auto rd = [&](int s) {
        av[s % (PFD + 1)] = *reinterpret_cast<const f32x4*>(afrag + s * AST);
        bv[s % (PFD + 1)] = *reinterpret_cast<const f32x2*>(bfrag + s * BST);
    };
#pragma unroll
    for (int s = 0; s < PFD && s < NS; ++s) rd(s);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + PFD < NS) rd(s + PFD);
        __builtin_amdgcn_sched_barrier(0);
        const int xi = xi_of(s);
        const f32x4 a = av[s % (PFD + 1)];
        const f32x2 b = bv[s % (PFD + 1)];
        acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[xi][0], 0, 0, 0);
        acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.x, acc[xi][1], 0, 0, 0);
        acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.y, acc[xi][0], 0, 0, 0);
        acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.y, acc[xi][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace
