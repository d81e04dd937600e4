// simt_macros.hpp -- the GPU intrinsics the per-wave kernel bodies (estep_block.hpp, gibbs_block.hpp) use, behind macros: in the
// product they expand to the intrinsic itself (same tokens, same code); tests/simt_emu.hpp defines them as exchanges
// through memory between one OS thread per lane, so that the same source runs on the CPU (tests/*_emu.cpp).
#pragma once
#ifndef RSEM_EMU
#define RSEM_DEVFN __device__ inline
#define RSEM_TIDX ((int)threadIdx.x)
#define RSEM_BDIM ((int)blockDim.x)
#define RSEM_SYNC() __syncthreads()
#define RSEM_SHFL_XOR(v, d) __shfl_xor(v, d)
#define RSEM_SHFL_DOWN(v, d) __shfl_down(v, d)
#define RSEM_SHFL_UP(v, d) __shfl_up(v, d)
#define RSEM_SHFL(v, src) __shfl(v, src)
#define RSEM_BALLOT(p) __ballot(p)
#define RSEM_READLANE(v, src) __builtin_amdgcn_readlane(v, src)
#define RSEM_ATOMIC_ADD(p, v) unsafeAtomicAdd(p, v)
#define RSEM_ATOMIC_ADD_I32(p, v) atomicAdd(p, v)
#define RSEM_LDS_ADD(p, v) (void)__builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)(p), v)
#define RSEM_LDS_ADD_I32(p, v) (void)__hip_atomic_fetch_add((__attribute__((address_space(3))) int*)(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
/* ... the returning form (ds_add_rtn_u32): a lane takes a place in a queue of its wave */
#define RSEM_LDS_FETCH_ADD_I32(p, v) __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
/* the lanes of a wave have all passed this point (the LDS serves a wave's instructions in order: what any lane wrote to LDS before it
   is there for every lane after it) */
#define RSEM_WAVE_SYNC()                                        \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)
/* every global load, store and atomic this wave has issued is done -- and the compiler knows it: no "possibly in flight" behind this */
#define RSEM_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70) /* vmcnt(0), expcnt and lgkmcnt untouched (gfx9 encoding: vmcnt = bits 15:14 | 3:0) */
#define RSEM_READFIRSTLANE(v) __builtin_amdgcn_readfirstlane(v)
/* the value must be in its register HERE: the wait for its load is placed at this point and not at a later join of paths */
#define RSEM_PIN(x) asm volatile("" : "+v"(x))
/* nothing moves across this point in the instruction schedule */
#define RSEM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define RSEM_RCP(x) __builtin_amdgcn_rcp(x)
#define RSEM_DPP_MOV(v, ctrl) __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, false)
#define RSEM_LL_AS_DOUBLE(x) __longlong_as_double(x)
#define RSEM_DOUBLE_AS_LL(x) __double_as_longlong(x)
/* a store of a value every storing lane agrees on: a plain store on the GPU; a relaxed atomic in the emulator, so that its
   ThreadSanitizer runs report exactly the accesses that are not meant to overlap */
#define RSEM_STORE_SAME(p, v) (*(p) = (v))
#define RSEM_NT_LOAD(p) __builtin_nontemporal_load(p)  /* read-once streams: kept out of the way of theta / counts in L2 and MALL */
#endif
