// Shared pieces of the persistent (one launch for all time steps) recurrent kernels: group / slot geometry, the workspace layout,
// the flag hand-off helpers and the XCD detection.  Protocol and measurements: gru_persist.hip (header comment), DESIGN.md section 4.
#pragma once

#include "common.h"

#define GT 16
#define GP_GROUPS 8
#define GP_SLOTS 32                          // flag words per group (H <= 512)
#define GP_KSTEPS 4                          // H / 128 MFMA k-steps per wave, H <= 512
#define GP_SPIN_LIMIT (1u << 22)              // polls before a wait gives up (a poll is 0.2-0.5 us: 1-2 s)
#define GP_TICKET_OFFSET (2 * GP_GROUPS * GP_SLOTS) // slot tickets, one counter per group (gp_claim_slot)
#define GP_FLAG_WORDS (GP_TICKET_OFFSET + GP_GROUPS) // per launch (zeroed by a memset node): step flags [8][32], XCC ids [8][32], tickets [8]
#define GP_SYNC_WORDS (GP_FLAG_WORDS + 4)          // + {sticky status, 3 pad}: 2096 bytes, a multiple of 16

typedef __bf16 gbf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ gbf8 as_bf8(u32x4 v) {
    union { u32x4 u; gbf8 b; } c;
    c.u = v;
    return c.b;
}

// the slot's flag: written through (sc1) in general; within one XCD a store that stays in the shared L2 (workgroup scope = no
// sc1 bit; the pollers' sc1 loads are L2-served)
__device__ __forceinline__ void gp_store_flag(gu32* flag, unsigned epoch, int one_xcd) {
    if (one_xcd)
        __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
        __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave 0: wait until every slot's flag of the group has reached `epoch`.  Returns false after GP_SPIN_LIMIT polls.
__device__ __forceinline__ bool gp_wait_flags(gu32* flags, int n_slots, unsigned epoch, int lane) {
    for (unsigned spins = 0;; ++spins) {
        const unsigned f = lane < n_slots ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
        if (__all(f >= epoch)) return true;
        if (spins > GP_SPIN_LIMIT) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// Workgroup barrier for LDS hand-overs inside the step loops: waits for this wave's LDS operations only.  __syncthreads() also
// drains the wave's global stores (its release fence = s_waitcnt vmcnt(0)): the waves that write a step's fp32 results would
// then reach the next barrier a store round trip late, every step.
__device__ __forceinline__ void gp_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Which (group, slot) a workgroup serves.  HIP promises nothing about the block -> XCD map (MI355X_MICROARCH.md: observed round-robin
// dealing, "for speed only"), and a group whose slots sit on several XCDs pays the write-through hand-off on every one of its T
// steps (C4 5.5 instead of 3.4 ms).  So membership is not read off the block index: the workgroup reads the id of the XCD it finds
// itself on and takes a ticket of THAT id's group - as long as the launch puts n_slots workgroups on every XCD (one per CU: the
// usual case) every group is complete on one XCD whatever the dealing order was.  A workgroup whose XCD's group is already full
// (an over-subscribed XCD) takes a ticket of the next group that still has a slot: 8 groups x n_slots slots, exactly as many
// workgroups, every ticket below n_slots is a distinct slot - so every workgroup finds one within 8 atomics and every slot gets
// an owner (if all 8 groups were full, 8 n_slots OTHER workgroups would own slots: one more than exist).  A group that ends up
// with members on several XCDs is found by gp_group_on_one_xcd below and keeps the placement-independent write-through form.
// `legacy` != 0 (MG_TUNE_GRU_HANDOFF bit 1, for A/B): group = block % 8 as before.
// Returns group * GP_SLOTS + slot, or -1 for a workgroup beyond the 8 n_slots the launch needs.  The tickets are zeroed with the
// flags by the memset node ahead of the launch.
__device__ __forceinline__ int gp_claim_slot(gu32* tickets, int n_slots, int tid, int* s_word, int legacy) {
    if (legacy) {
        const int slot = blockIdx.x / GP_GROUPS;
        return slot < n_slots ? (int)(blockIdx.x % GP_GROUPS) * GP_SLOTS + slot : -1;
    }
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(id));
        int got = -1;
        for (int a = 0; a < GP_GROUPS && got < 0; ++a) {
            const int g = (int)((id + (unsigned)a) % GP_GROUPS);
            const unsigned t = __hip_atomic_fetch_add(tickets + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < (unsigned)n_slots) got = g * GP_SLOTS + (int)t;
        }
        *s_word = got;
    }
    __syncthreads();
    const int v = *s_word;
    __syncthreads();                                   // the caller reuses the word
    return v;
}

// Where the group runs.  Every workgroup publishes the id of the XCD it is on (s_getreg HW_REG_XCC_ID) and reads the ids of the
// group's other slots - once per launch, with the placement-independent sc1 protocol.  If they are all equal, the whole group
// shares one L2: the hand-off stores may then stay plain (the line stays in that L2, where the readers' sc1 loads - which only
// bypass their own CU's L1 - find it) instead of being written through to memory and fetched back over the fabric.  The
// block -> XCD map itself is never assumed: a group spread over several XCDs keeps the write-through form.
// Returns 1 = one XCD, 0 = several, -1 = timed out.
__device__ __forceinline__ int gp_group_on_one_xcd(gu32* xcc_tab, int slot, int n_slots, int tid, int* s_word) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(id));
    if (tid == 0) __hip_atomic_store(xcc_tab + slot, id + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
        int result = -1;
        for (unsigned spins = 0; spins <= GP_SPIN_LIMIT; ++spins) {
            const unsigned v = tid < n_slots ? __hip_atomic_load(xcc_tab + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : id + 1u;
            if (__all(v != 0u)) {
                result = __all(v == id + 1u) ? 1 : 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (tid == 0) *s_word = result;
    }
    __syncthreads();
    return *s_word;
}

// Residency: every workgroup of a persistent launch must be on the device at once.  The kernels need up to 256 VGPRs, i.e. as
// few as two 256-thread workgroups per CU: a launch is offered only if the current device has room for `workgroups` at two
// per CU (MI355X: 256 CUs; a partitioned or harvested device simply takes the per-step kernels instead of timing out).
static inline int gp_device_holds(long workgroups) {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cus = n;
    }
    return cus > 0 && workgroups <= 2L * cus;
}

// workspace: [step flags 8 x 32 words | XCC ids 8 x 32 words | status word + 3 pad] [hand-off ring]
#define GP_RING_OFFSET ((size_t)GP_SYNC_WORDS * sizeof(unsigned))
