// Microbenchmark: latency of the memory operations workgroups of one kernel can use to talk to each other on MI355X --
// dependent chains of loads / atomics at workgroup, agent and system scope, and a ping-pong between two workgroups on the same
// XCD and on different XCDs.  Decides what the team barrier and the cross look-ups of gibbs_exact_team.hpp cost per round trip.
// Build: hipcc --offload-arch=gfx950 -O3 latency.hip -o latency ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ inline unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int MODE>
__global__ void k_chase(unsigned long long* next, int iters, unsigned long long* out) {
    if (threadIdx.x != 0) return;
    unsigned long long i = 0;
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < iters; k++) {
        if (MODE == 0) i = next[i];
        if (MODE == 1) i = __hip_atomic_load(&next[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 2) i = __hip_atomic_load(&next[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 3) i = __hip_atomic_load(&next[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (MODE == 4) i = __hip_atomic_fetch_add(&next[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 5) i = __hip_atomic_fetch_add(&next[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 6) { unsigned long long v; asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(&next[i]) : "memory"); i = v; }
        if (MODE == 7) { unsigned long long v; asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(&next[i]) : "memory"); i = v; }
        if (MODE == 8) { unsigned long long v; asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(&next[i]) : "memory"); i = v; }
    }
    const unsigned long long t1 = wall_clock64();
    out[0] = t1 - t0;
    out[1] = i;
}

// ping-pong: block A (blockIdx a) and block B (blockIdx b) bounce a counter `iters` times.  SCOPE 0: agent-scope atomics,
// 1: workgroup-scope atomic RMWs (performed in the XCD's L2) both for writing and for reading (fetch_add 0),
// 2: workgroup-scope RMW writes, sc0 loads for reading.
template <int SCOPE>
__global__ void k_pingpong(int a, int b, unsigned long long* flag, int iters, unsigned long long* out, unsigned* xcc) {
    if (threadIdx.x != 0) return;
    const int me = (int)blockIdx.x == a ? 0 : ((int)blockIdx.x == b ? 1 : -1);
    if (me < 0) return;
    xcc[me] = xcc_id();
    auto rd = [&]() -> unsigned long long {
        if (SCOPE == 0) return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (SCOPE == 1) return __hip_atomic_fetch_add(flag, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned long long v;
        asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(flag) : "memory");
        return v;
    };
    auto bump = [&]() {
        if (SCOPE == 0) (void)__hip_atomic_fetch_add(flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else (void)__hip_atomic_fetch_add(flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    const unsigned long long t0 = wall_clock64();
    unsigned long long spins = 0;
    bool ok = true;
    for (int k = 0; k < iters && ok; k++) {
        const unsigned long long want = 2ull * k + (unsigned long long)me;  // A bumps at even values, B at odd
        unsigned long long s = 0;
        while (rd() != want) { if (++s > 2000000ull) { ok = false; break; } }
        spins += s;
        if (ok) bump();
    }
    const unsigned long long t1 = wall_clock64();
    out[me * 3 + 0] = t1 - t0;
    out[me * 3 + 1] = spins;
    out[me * 3 + 2] = ok ? 1 : 0;
}

int main() {
    const int N = 4096, iters = 4000;
    std::vector<unsigned long long> h(N);
    // a random cycle over N entries (stride pattern that defeats any prefetch)
    std::vector<int> perm(N);
    for (int i = 0; i < N; i++) perm[i] = i;
    srand(1);
    for (int i = N - 1; i > 0; i--) { int j = rand() % (i + 1); int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
    for (int i = 0; i < N; i++) h[perm[i]] = perm[(i + 1) % N];
    unsigned long long *d, *out;
    unsigned* xcc;
    CK(hipMalloc(&d, sizeof(unsigned long long) * N));
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&xcc, 16));
    CK(hipMemcpy(d, h.data(), sizeof(unsigned long long) * N, hipMemcpyHostToDevice));
    const char* names[] = {"plain load", "atomic load workgroup", "atomic load agent", "atomic load system", "fetch_add(0) agent", "fetch_add(0) workgroup",
                           "asm load sc0", "asm load sc1", "asm load nt"};
    unsigned long long ho[8];
#define RUN(M) for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_chase<M>, dim3(1), dim3(64), 0, 0, d, iters, out); CK(hipDeviceSynchronize()); } \
    CK(hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost)); printf("chain of %-26s %7.1f ns per op\n", names[M], (double)ho[0] * 10.0 / iters);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
    unsigned long long* flag;
    CK(hipMalloc(&flag, 8));
    unsigned hx[2];
    const int pairs[3][2] = {{0, 8}, {0, 1}, {0, 4}};
#define PP(S, label) for (int p = 0; p < 3; p++) { CK(hipMemset(flag, 0, 8)); CK(hipMemset(out, 0, 64)); \
        hipLaunchKernelGGL(k_pingpong<S>, dim3(16), dim3(64), 0, 0, pairs[p][0], pairs[p][1], flag, 2000, out, xcc); CK(hipDeviceSynchronize()); \
        CK(hipMemcpy(ho, out, 48, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost)); \
        printf("ping-pong %-34s blocks %d,%d (XCC %u,%u): %7.1f ns per hand-over, %s\n", label, pairs[p][0], pairs[p][1], hx[0], hx[1], \
               (double)ho[0] * 10.0 / 4000.0, (ho[2] && ho[5]) ? "completed" : "GAVE UP (the other side's writes were not seen)"); }
    PP(0, "agent-scope atomics")
    PP(1, "workgroup-scope RMW both ways")
    PP(2, "workgroup-scope RMW + sc0 loads")
    return 0;
}
