// What a hand-over INSIDE a launch costs on MI355X against a kernel boundary (fold k_scan into the tail of
// k_rwseg -- the last workgroup of a group to finish composes the group's maps -- and the persistent form's grid barrier).
//
//   produce          256 workgroups x 1024 threads, each writes its 2 KB "segment map" (stands for k_rwseg's tail)
//   compose          16 workgroups: 16 maps -> LDS, 16 dependent lookups per state, group map + 16 prefix maps out (k_scan)
//   produce+compose  as two launches of one stream, back to back                                   (today's flow)
//   fused            produce; __threadfence(); ticket = atomicAdd(group counter); the workgroup that draws the last ticket of
//                    its group does the compose (acquire fence first), nobody waits                (item 3a)
//   grid barrier     256 resident workgroups: arrive (release + atomic), spin until all arrived (acquire)  (item 3b's kill criterion)
//
// hipcc --offload-arch=gfx950 -O3 -o ubench_handover scratch/ubench_handover.hip && ./ubench_handover
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define NS 1024
#define S 256
#define G2 16

__global__ void __launch_bounds__(1024) k_produce(uint16_t *maps, int salt)
{
    maps[(size_t)blockIdx.x * NS + threadIdx.x] = (uint16_t)((threadIdx.x * 7 + blockIdx.x + salt) & (NS - 1));
}

__device__ __forceinline__ void compose_group(const uint16_t *maps, uint16_t *pmaps, uint16_t *gmaps, int grp, uint16_t *M)
{
    const uint4 *s4 = reinterpret_cast<const uint4 *>(maps + (size_t)grp * G2 * NS);
    uint4 *d4 = reinterpret_cast<uint4 *>(M);
    for (int e = threadIdx.x; e < G2 * NS / 8; e += 1024) d4[e] = s4[e];
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
    compose_group(maps, pmaps, gmaps, blockIdx.x, M);
}

__global__ void __launch_bounds__(1024) k_fused(uint16_t *maps, uint16_t *pmaps, uint16_t *gmaps, unsigned *tickets, int salt, unsigned round)
{
    __shared__ __align__(16) uint16_t M[G2 * NS];
    __shared__ unsigned s_ticket;
    maps[(size_t)blockIdx.x * NS + threadIdx.x] = (uint16_t)((threadIdx.x * 7 + blockIdx.x + salt) & (NS - 1));
    __syncthreads();                                     // every thread's store is issued ...
    const int grp = blockIdx.x / G2;
    if (threadIdx.x == 0) {
        __threadfence();                                 // ... and released at device scope (an L2 write-back on an 8-XCD part)
        s_ticket = atomicAdd(&tickets[grp], 1u);
    }
    __syncthreads();
    if (s_ticket != round * G2 + (G2 - 1)) return;       // (tickets keep counting from launch to launch: no reset kernel)
    __threadfence();                                     // acquire: the other XCDs' maps
    compose_group(maps, pmaps, gmaps, grp, M);
}

__global__ void __launch_bounds__(1024) k_barrier(unsigned *counter, unsigned *out, int nbar, unsigned base)
{
    unsigned long long t0 = 0;
    for (int b = 0; b < nbar; b++) {
        __syncthreads();
        if (threadIdx.x == 0) {
            if (b == 1) t0 = __builtin_amdgcn_s_memrealtime();
            __threadfence();
            atomicAdd(counter, 1u);
            const unsigned want = base + (unsigned)(b + 1) * gridDim.x;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
            __threadfence();
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

int main()
{
    uint16_t *maps, *pmaps, *gmaps, *pm2, *gm2;
    unsigned *tickets, *counter, *out;
    hipMalloc(&maps, S * NS * 2); hipMalloc(&pmaps, S * NS * 2); hipMalloc(&gmaps, (S / G2) * NS * 2);
    hipMalloc(&pm2, S * NS * 2); hipMalloc(&gm2, (S / G2) * NS * 2);
    hipMalloc(&tickets, 64 * 4); hipMalloc(&counter, 4); hipMalloc(&out, 4);
    hipMemset(tickets, 0, 64 * 4); hipMemset(counter, 0, 4);
    const int reps = 2000;
    const float t_prod = timed([&](int i) { hipLaunchKernelGGL(k_produce, dim3(S), dim3(1024), 0, 0, maps, i); }, reps);
    const float t_two = timed([&](int i) {
        hipLaunchKernelGGL(k_produce, dim3(S), dim3(1024), 0, 0, maps, i);
        hipLaunchKernelGGL(k_compose, dim3(S / G2), dim3(1024), 0, 0, (const uint16_t *)maps, pmaps, gmaps);
    }, reps);
    unsigned round = 0;
    const float t_fused = timed([&](int i) { hipLaunchKernelGGL(k_fused, dim3(S), dim3(1024), 0, 0, maps, pm2, gm2, tickets, i, round++); }, reps);
    // same answers either way (the last iteration of each used the same salt: 20 + reps - 1)
    std::vector<uint16_t> a((S / G2) * NS), b((S / G2) * NS), c(S * NS), d(S * NS);
    hipMemcpy(a.data(), gmaps, a.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), gm2, b.size() * 2, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), pmaps, c.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(d.data(), pm2, d.size() * 2, hipMemcpyDeviceToHost);
    bool same = a == b;
    for (int s = 0; s < S && same; s++)
        if (s % G2) for (int x = 0; x < NS; x++) if (c[(size_t)s * NS + x] != d[(size_t)s * NS + x]) { same = false; break; }
    // grid barrier: 256 workgroups (one per CU), 201 barriers, the first one not timed
    unsigned base = 0;
    hipLaunchKernelGGL(k_barrier, dim3(256), dim3(1024), 0, 0, counter, out, 201, base);
    hipDeviceSynchronize();
    unsigned ticks = 0;
    hipMemcpy(&ticks, out, 4, hipMemcpyDeviceToHost);
    printf("produce alone (256 workgroups, 2 KB each)              %6.2f us per launch\n", t_prod);
    printf("produce + compose, two launches back to back           %6.2f us per pair   (+%.2f us for the boundary and k_scan's work)\n", t_two, t_two - t_prod);
    printf("fused: last workgroup of a group composes (release/acquire) %6.2f us per launch (+%.2f us)   results %s\n", t_fused, t_fused - t_prod, same ? "identical" : "DIFFER");
    printf("grid barrier over 256 resident workgroups              %6.2f us per barrier (200 barriers, s_memrealtime)\n", ticks / 100.0 / 200.0);
    return same ? 0 : 1;
}
