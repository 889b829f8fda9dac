// lds_dma.h -- global memory -> LDS transfers (16 bytes per lane) shared by the scan and the encoder kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace crs {
namespace {

// A pointer that IS wave-uniform (kernel argument + blockIdx arithmetic), made provably so for the
// "s" (SGPR) operand of an inline-asm load: both halves go through v_readfirstlane.
// HAZARD: an SGPR written by v_readfirstlane needs 5 wait states before a VMEM instruction may read
// it as its base, and hipcc pads nothing inside an asm string -- every asm load that takes such a
// pointer therefore opens with "s_nop 4" (cdna_hip_programming.md section 5.7 item 2).
template <typename T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return reinterpret_cast<const T*>(((unsigned long long)hi << 32) | lo);
}

// ---------------------------------------------------------------- LDS-DMA (global memory -> LDS, 16 bytes per lane)
// global_load_lds_dwordx4 takes its LDS destination (wave-uniform base; lane i lands at base + 16 i) from M0.  The
// statement is asm because the compiler's own waitcnt insertion would put a vmcnt(0) in front of every LDS read that
// may alias an in-flight transfer -- the kernels count their transfers themselves.  M0 is SAVED AND RESTORED inside the
// statement (no reserved register in a clobber list, nothing the compiler keeps in M0 is disturbed).  Wait states:
// the two s_mov + s_nop 2 are the 5 states a v_readfirstlane-written SGPR base needs before a VMEM read of it, and
// cover the 1 state between the write of M0 and the transfer.
__device__ __forceinline__ void lds_dma16(unsigned lds_dst, unsigned voff, const void* sbase) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void lds_dma16(unsigned lds_dst, const void* vaddr) {   // per-lane 64-bit address
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_dst), "v"(vaddr) : "memory");
}

}  // namespace
}  // namespace crs
