// Diagnostic builds only (-DCCV_DIAG: `python tools/ablate.py stamp=-DCCV_DIAG`, read back by tools/stamps_r4.py).  Never part
// of the product library: mppi_kernels.h includes this file under CCV_DIAG and defines the hook as nothing otherwise.
//
// CCV_DIAG_STAMP(A, slot): the 100 MHz wall clock (s_memrealtime) into slot `slot` (0 .. kDiagSlots-1) of this workgroup's row
// of the stamp buffer RolloutArgs::dbg ([kDiagHeader + workgroup * kDiagSlots + slot], the first kDiagBlocks workgroups
// of the grid).  Slots of the four-wave kernel (mppi_rollout_r4.h):
//   0 entry   1 first barrier passed   2 staging barrier passed   3 normals of block 0 published   4 the dynamics wave has them
//   5 block 0 published by the dynamics wave   6 dynamics loop end   7 distance loop end   8 wave 0 past the barrier
//   9 wave 0 has its weight   10 / 12 / 13 wave 0 / 2 / 3 through the epilogue   11 wave 2 has its weight
//   14 the distance wave is through block 0   15 noise loop end
#pragma once
#if !defined(CCV_DIAG)
#error "mppi_diag.h belongs to -DCCV_DIAG builds only"
#endif

namespace ccv {
constexpr int kDiagHeader = 64, kDiagSlots = 16, kDiagBlocks = 4096;
}
#define CCV_DIAG_STAMP(A, slot)                                                                                              \
    do {                                                                                                                     \
        if ((A).dbg && blockIdx.x < ccv::kDiagBlocks)                                                                        \
            (A).dbg[ccv::kDiagHeader + blockIdx.x * ccv::kDiagSlots + (slot)] = __builtin_amdgcn_s_memrealtime();             \
    } while (0)
// ... once `value` has been computed (the stamp is not scheduled ahead of it)
#define CCV_DIAG_STAMP_VALUE(A, slot, value)                                                                                 \
    do {                                                                                                                     \
        double keep__ = (value);                                                                                             \
        asm volatile("" : "+v"(keep__));                                                                                     \
        CCV_DIAG_STAMP(A, slot);                                                                                             \
    } while (0)
