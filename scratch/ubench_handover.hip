// What a hand-over INSIDE a launch costs on MI355X against a kernel boundary (fold k_scan into the
// tail of k_rwseg -- the last workgroup of a group to finish composes the group's maps -- and the persistent form's grid barrier).
//
//   produce          256 workgroups x 1024 threads, each writes its 2 KB "segment map" (stands for k_rwseg's tail); `work` dependent
//                    ALU iterations in front of the store stand for the kernel body (0: nothing; ~17 us: k_rwseg's)
//   compose          16 workgroups: 16 maps -> LDS, 16 dependent lookups per state, group map + 16 prefix maps out (k_scan)
//   produce+compose  as two launches of one stream, back to back                                   (today's flow)
//   fused/fence      produce; __threadfence(); ticket = atomicAdd(group counter); the workgroup that draws the last ticket of
//                    its group does the compose (acquire fence first), nobody waits                (round 5: +28 us)
//   fused/wt16       NO fence: the map leaves as 16-byte WRITE-THROUGH stores (buffer_store_dwordx4 sc1: 128 lanes), every storing
//                    wave drains (s_waitcnt vmcnt(0)), barrier, one relaxed agent-scope ticket; the last arriver reads the 16
//                    maps with 16-byte sc1 loads (they bypass its L1; per-XCD L2s hold no dirty copy of a written-through line)
//   fused/wt8        the same with 8-byte agent-scope relaxed atomics on both sides (global_store/load_dwordx2 sc1)
//   fused/uc         the maps in hipDeviceMallocUncached memory, plain stores and loads, drain + ticket, no fence
//   grid barrier     256 resident workgroups: fence form (release + atomic, spin, acquire) and the fence-free form (relaxed
//                    agent add, sc1 poll) flat and XCD-hierarchical                                 (item 3b's kill criterion)
// Every fused variant is run with the consumer's L1 WARM (each workgroup pre-reads its group's maps with plain loads before it
// produces: a stale line would be served from L1) and under UNEVEN load (workgroup b works (b % 5) times `work`), and every word
// of the result is compared with the two-launch form's.
//
// hipcc --offload-arch=gfx950 -O3 -o ubench_handover scratch/ubench_handover.hip && ./ubench_handover
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define NS 1024
#define S 256
#define G2 16

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// the kernel body's stand-in: `n` dependent multiply-adds per thread (4 cycles each on a SIMD: n = 10 000 is ~17 us)
__device__ __forceinline__ unsigned body(unsigned x, int n)
{
    for (int i = 0; i < n; i++) x = x * 1664525u + 1013904223u;
    return x;
}
__device__ __forceinline__ uint16_t map_value(int salt, unsigned junk)
{
    return (uint16_t)(((threadIdx.x * 7 + blockIdx.x + salt) & (NS - 1)) | (junk == 0x12345u ? 1u : 0u));      // (junk never matches: keeps the body alive)
}

__global__ void __launch_bounds__(1024) k_produce(uint16_t *maps, int salt, int work, int uneven)
{
    const unsigned j = body(threadIdx.x + salt, uneven ? work * (int)(blockIdx.x % 5) : work);
    maps[(size_t)blockIdx.x * NS + threadIdx.x] = map_value(salt, j);
}

// MODE 0: plain 16-byte loads; 1: 16-byte sc1 buffer loads; 2: 8-byte agent-scope relaxed loads
template <int MODE>
__device__ __forceinline__ void compose_group(const uint16_t *maps, uint16_t *pmaps, uint16_t *gmaps, int grp, uint16_t *M)
{
    const uint16_t *src = maps + (size_t)grp * G2 * NS;
    if constexpr (MODE == 0) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(M);
        for (int e = threadIdx.x; e < G2 * NS / 8; e += 1024) d4[e] = s4[e];
    } else if constexpr (MODE == 1) {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(src), 0, G2 * NS * 2, 0x00020000);
        u32x4 *d4 = reinterpret_cast<u32x4 *>(M);
        u32x4 v[G2 * NS / 8 / 1024];
#pragma unroll
        for (int k = 0; k < G2 * NS / 8 / 1024; k++) v[k] = __builtin_amdgcn_raw_buffer_load_b128(r, (threadIdx.x + k * 1024) * 16, 0, 16);     // aux 16 = sc1
#pragma unroll
        for (int k = 0; k < G2 * NS / 8 / 1024; k++) d4[threadIdx.x + k * 1024] = v[k];
    } else {
        const unsigned long long *s8 = reinterpret_cast<const unsigned long long *>(src);
        unsigned long long *d8 = reinterpret_cast<unsigned long long *>(M);
        unsigned long long v[G2 * NS / 4 / 1024];
#pragma unroll
        for (int k = 0; k < G2 * NS / 4 / 1024; k++) v[k] = __hip_atomic_load(s8 + threadIdx.x + k * 1024, RLX_AGENT);
#pragma unroll
        for (int k = 0; k < G2 * NS / 4 / 1024; k++) d8[threadIdx.x + k * 1024] = v[k];
    }
    __syncthreads();
    int x = threadIdx.x;
#pragma unroll
    for (int j = 0; j < G2; j++) {
        if (j > 0) pmaps[(size_t)(grp * G2 + j) * NS + threadIdx.x] = (uint16_t)x;
        x = M[j * NS + x];
    }
    gmaps[(size_t)grp * NS + threadIdx.x] = (uint16_t)x;
}

__global__ void __launch_bounds__(1024) k_compose(const uint16_t *maps, uint16_t *pmaps, uint16_t *gmaps)
{
    __shared__ __align__(16) uint16_t M[G2 * NS];
    compose_group<0>(maps, pmaps, gmaps, blockIdx.x, M);
}

// every workgroup reads its group's maps with PLAIN loads first: the lines sit in this CU's L1 when the hand-over comes
__device__ __forceinline__ unsigned warm_l1(const uint16_t *maps, int grp)
{
    const uint4 *s4 = reinterpret_cast<const uint4 *>(maps + (size_t)grp * G2 * NS);
    unsigned acc = 0;
    for (int e = threadIdx.x; e < G2 * NS / 8; e += 1024) { const uint4 v = s4[e]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    return acc;
}

// VAR 0: fences (round 5);  1: 16-byte write-through stores + sc1 loads;  2: 8-byte agent atomics;  3: plain stores / loads (uncached memory)
template <int VAR>
__global__ void __launch_bounds__(1024) k_fused(uint16_t *maps, uint16_t *pmaps, uint16_t *gmaps, unsigned *tickets, int salt, unsigned round,
                                                 int work, int uneven, int warm, unsigned *sink)
{
    __shared__ __align__(16) uint16_t M[G2 * NS];
    __shared__ unsigned s_ticket;
    const int grp = blockIdx.x / G2;
    unsigned junk = 0;
    if (warm) junk = warm_l1(maps, grp);
    if (junk == 0xdeadbeefu) sink[0] = junk;
    const unsigned j = body(threadIdx.x + salt, uneven ? work * (int)(blockIdx.x % 5) : work);
    const uint16_t val = map_value(salt, j);
    if constexpr (VAR == 0 || VAR == 3) {
        maps[(size_t)blockIdx.x * NS + threadIdx.x] = val;
        if (VAR == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains
        __syncthreads();
        if (threadIdx.x == 0) {
            if (VAR == 0) __threadfence();                                   // released at device scope (an L2 write-back on an 8-XCD part)
            s_ticket = VAR == 0 ? atomicAdd(&tickets[grp], 1u) : __hip_atomic_fetch_add(&tickets[grp], 1u, RLX_AGENT);
        }
    } else {
        M[threadIdx.x] = val;                                                // pack the map into 16- / 8-byte units through LDS
        __syncthreads();
        if constexpr (VAR == 1) {
            if (threadIdx.x < NS / 8) {
                __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(maps + (size_t)blockIdx.x * NS, 0, NS * 2, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(reinterpret_cast<const u32x4 *>(M)[threadIdx.x], r, threadIdx.x * 16, 0, 16);     // sc1: write-through
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            if (threadIdx.x < NS / 4) {
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(maps + (size_t)blockIdx.x * NS) + threadIdx.x,
                                   reinterpret_cast<const unsigned long long *>(M)[threadIdx.x], RLX_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();                                                     // the storing waves have drained
        if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(&tickets[grp], 1u, RLX_AGENT);
    }
    __syncthreads();
    if (s_ticket != round * G2 + (G2 - 1)) return;       // (tickets keep counting from launch to launch: no reset kernel)
    if constexpr (VAR == 0) { __threadfence(); compose_group<0>(maps, pmaps, gmaps, grp, M); }
    else if constexpr (VAR == 1) compose_group<1>(maps, pmaps, gmaps, grp, M);
    else if constexpr (VAR == 2) compose_group<2>(maps, pmaps, gmaps, grp, M);
    else compose_group<0>(maps, pmaps, gmaps, grp, M);
}

// grid barriers.  KIND 0: fences around a flat counter (round 5); 1: no fence, relaxed agent add + sc1 poll, flat counter;
// 2: no fence, XCD-hierarchical (per-XCD arrival counter; the last arriver of an XCD adds to the top counter; everybody polls the
// top counter's generation)
template <int KIND>
__global__ void __launch_bounds__(1024) k_barrier(unsigned *counter, unsigned *out, int nbar, unsigned base)
{
    unsigned long long t0 = 0;
    unsigned xcc = 0;
    if (KIND == 2) asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    for (int b = 0; b < nbar; b++) {
        __syncthreads();
        if (threadIdx.x == 0) {
            if (b == 1) t0 = __builtin_amdgcn_s_memrealtime();
            if constexpr (KIND == 0) {
                __threadfence();
                atomicAdd(counter, 1u);
                const unsigned want = base + (unsigned)(b + 1) * gridDim.x;
                for (int sp = 0; sp < (1 << 22) && __hip_atomic_load(counter, RLX_AGENT) < want; sp++) __builtin_amdgcn_s_sleep(1);      // (bounded: never hang the box)
                __threadfence();
            } else if constexpr (KIND == 1) {
                __hip_atomic_fetch_add(counter, 1u, RLX_AGENT);
                const unsigned want = base + (unsigned)(b + 1) * gridDim.x;
                for (int sp = 0; sp < (1 << 22) && __hip_atomic_load(counter, RLX_AGENT) < want; sp++) __builtin_amdgcn_s_sleep(1);
            } else {
                // counter[16 * (1 + xcc)]: arrivals of this XCD (32 per barrier); counter[0]: XCDs through (8 per barrier)
                const unsigned per = gridDim.x / 8;
                const unsigned got = __hip_atomic_fetch_add(&counter[16 * (1 + xcc)], 1u, RLX_AGENT);
                if ((got + 1) % per == 0) __hip_atomic_fetch_add(&counter[0], 1u, RLX_AGENT);
                const unsigned want = base + (unsigned)(b + 1) * 8u;
                int sp = 0;
                for (; sp < (1 << 22) && __hip_atomic_load(&counter[0], RLX_AGENT) < want; sp++) __builtin_amdgcn_s_sleep(1);
                if (sp == (1 << 22)) out[1] = 1;        // (workgroups were not dealt 32 per XCD: the figure is void)
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (unsigned)(__builtin_amdgcn_s_memrealtime() - t0);     // 100 MHz ticks over nbar - 1 barriers
}

template <typename F> static float timed(F f, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; i++) f(i);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) f(20 + i);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

static uint16_t *maps, *pmaps, *gmaps, *pm2, *gm2, *maps_uc;
static unsigned *tickets, *counter, *out, *sink;

static bool same_as_reference(uint16_t *gm, uint16_t *pm)
{
    std::vector<uint16_t> a((S / G2) * NS), b((S / G2) * NS), c(S * NS), d(S * NS);
    hipMemcpy(a.data(), gmaps, a.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), gm, b.size() * 2, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), pmaps, c.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(d.data(), pm, d.size() * 2, hipMemcpyDeviceToHost);
    bool same = a == b;
    for (int s = 0; s < S && same; s++)
        if (s % G2) for (int x = 0; x < NS; x++) if (c[(size_t)s * NS + x] != d[(size_t)s * NS + x]) { same = false; break; }
    return same;
}

template <int VAR> static void run_fused(const char *name, uint16_t *mp, int work, int uneven, int warm, float t_prod, int reps)
{
    hipMemset(tickets, 0, 64 * 4);
    hipMemset(pm2, 0xff, S * NS * 2); hipMemset(gm2, 0xff, (S / G2) * NS * 2);
    unsigned round = 0;
    // (every iteration is checked against the reference of its salt only at the end: the last iteration of each form used salt 20 + reps - 1)
    const float t = timed([&](int i) { hipLaunchKernelGGL(k_fused<VAR>, dim3(S), dim3(1024), 0, 0, mp, pm2, gm2, tickets, i, round++, work, uneven, warm, sink); }, reps);
    hipDeviceSynchronize();
    const bool same = same_as_reference(gm2, pm2);
    printf("  fused/%-6s work %5d %s %s  %7.2f us per launch (%+.2f us over produce alone)   results %s\n", name, work, uneven ? "uneven" : "even  ",
           warm ? "L1 warm" : "L1 cold", t, t - t_prod, same ? "identical" : "DIFFER");
}

int main()
{
    hipMalloc(&maps, S * NS * 2); hipMalloc(&pmaps, S * NS * 2); hipMalloc(&gmaps, (S / G2) * NS * 2);
    hipMalloc(&pm2, S * NS * 2); hipMalloc(&gm2, (S / G2) * NS * 2);
    hipMalloc(&tickets, 64 * 4); hipMalloc(&counter, 4096); hipMalloc(&out, 8); hipMemset(out, 0, 8); hipMalloc(&sink, 4);
    if (hipExtMallocWithFlags((void **)&maps_uc, S * NS * 2, hipDeviceMallocUncached) != hipSuccess) { maps_uc = nullptr; (void)hipGetLastError(); }
    hipMemset(tickets, 0, 64 * 4); hipMemset(counter, 0, 4096);
    const int reps = 1000;
    for (int work : {0, 3000}) {
        for (int uneven : {0, 1}) {
            if (work == 0 && uneven) continue;
            const float t_prod = timed([&](int i) { hipLaunchKernelGGL(k_produce, dim3(S), dim3(1024), 0, 0, maps, i, work, uneven); }, reps);
            const float t_two = timed([&](int i) {
                hipLaunchKernelGGL(k_produce, dim3(S), dim3(1024), 0, 0, maps, i, work, uneven);
                hipLaunchKernelGGL(k_compose, dim3(S / G2), dim3(1024), 0, 0, (const uint16_t *)maps, pmaps, gmaps);
            }, reps);
            hipDeviceSynchronize();
            printf("work %5d %s: produce alone %7.2f us per launch; produce + compose as two launches %7.2f us (+%.2f us for the boundary and k_scan's work)\n",
                   work, uneven ? "uneven" : "even  ", t_prod, t_two, t_two - t_prod);
            for (int warm : {0, 1}) {
                run_fused<0>("fence", maps, work, uneven, warm, t_prod, reps);
                run_fused<1>("wt16", maps, work, uneven, warm, t_prod, reps);
                run_fused<2>("wt8", maps, work, uneven, warm, t_prod, reps);
                if (maps_uc) run_fused<3>("uc", maps_uc, work, uneven, warm, t_prod, reps);
            }
        }
    }
    // a NEGATIVE control: plain stores, no fence, ordinary memory -- must be seen to differ at least sometimes, or the check above proves nothing
    run_fused<3>("PLAIN!", maps, 3000, 1, 1, 0.f, reps);
    // grid barriers: 256 workgroups (one per CU), 201 barriers, the first one not timed
    const char *names[3] = {"fences, flat counter (round 5)", "no fence: relaxed add + sc1 poll, flat counter", "no fence, XCD-hierarchical"};
    for (int kind = 0; kind < 3; kind++) {
        hipMemset(counter, 0, 4096);
        if (kind == 0) hipLaunchKernelGGL(k_barrier<0>, dim3(256), dim3(1024), 0, 0, counter, out, 201, 0u);
        else if (kind == 1) hipLaunchKernelGGL(k_barrier<1>, dim3(256), dim3(1024), 0, 0, counter, out, 201, 0u);
        else hipLaunchKernelGGL(k_barrier<2>, dim3(256), dim3(1024), 0, 0, counter, out, 201, 0u);
        hipDeviceSynchronize();
        unsigned ticks[2] = {0, 0};
        hipMemcpy(ticks, out, 8, hipMemcpyDeviceToHost);
        printf("grid barrier over 256 resident workgroups, %-48s %6.2f us per barrier (200 barriers, s_memrealtime)%s\n", names[kind], ticks[0] / 100.0 / 200.0,
               ticks[1] ? "  TIMED OUT: void" : "");
    }
    return 0;
}
