// lone-wave issue rates: independent instruction streams of each type, s_memtime around them
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 200
__global__ void k(double* out, unsigned long long* t, const int* idx, int mode)
{
    const int lane = threadIdx.x;
    double a0 = out[lane], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = out[lane + 64];
    int x0 = idx[0], x1 = idx[1], x2 = idx[2], x3 = idx[3], x4 = idx[4], x5 = idx[5], x6 = idx[6], x7 = idx[7];
    int v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3, v4 = lane + 4, v5 = lane + 5, v6 = lane + 6, v7 = lane + 7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {
        for (int i = 0; i < REP; i++)
            asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if (mode == 1) {
        for (int i = 0; i < REP; i++)
            asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 3\n s_add_u32 %3, %3, 3\n s_add_u32 %4, %4, 3\n s_add_u32 %5, %5, 3\n s_add_u32 %6, %6, 3\n s_add_u32 %7, %7, 3"
                         : "+s"(x0), "+s"(x1), "+s"(x2), "+s"(x3), "+s"(x4), "+s"(x5), "+s"(x6), "+s"(x7) :: "scc");
    } else if (mode == 2) {
        for (int i = 0; i < REP; i++)
            asm volatile("v_add_u32 %0, %0, 3\n v_add_u32 %1, %1, 3\n v_add_u32 %2, %2, 3\n v_add_u32 %3, %3, 3\n v_add_u32 %4, %4, 3\n v_add_u32 %5, %5, 3\n v_add_u32 %6, %6, 3\n v_add_u32 %7, %7, 3"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
    } else if (mode == 3) {   // alternating VALU f64 / SALU, all independent
        for (int i = 0; i < REP; i++)
            asm volatile("v_add_f64 %0, %0, %8\n s_add_u32 %9, %9, 3\n v_add_f64 %1, %1, %8\n s_add_u32 %10, %10, 3\n v_add_f64 %2, %2, %8\n s_add_u32 %11, %11, 3\n v_add_f64 %3, %3, %8\n s_add_u32 %12, %12, 3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) , "+v"(b), "+s"(x0), "+s"(x1), "+s"(x2), "+s"(x3) :: "scc");
    } else if (mode == 4) {   // dependent SALU chain
        for (int i = 0; i < REP; i++)
            asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3"
                         : "+s"(x0) :: "scc");
    } else if (mode == 5) {   // dependent VALU u32 chain
        for (int i = 0; i < REP; i++)
            asm volatile("v_add_u32 %0, %0, 3\n v_add_u32 %0, %0, 3\n v_add_u32 %0, %0, 3\n v_add_u32 %0, %0, 3\n v_add_u32 %0, %0, 3\n v_add_u32 %0, %0, 3\n v_add_u32 %0, %0, 3\n v_add_u32 %0, %0, 3"
                         : "+v"(v0));
    } else if (mode == 6) {   // independent v_max_f64
        for (int i = 0; i < REP; i++)
            asm volatile("v_max_f64 %0, %0, %8\n v_max_f64 %1, %1, %8\n v_max_f64 %2, %2, %8\n v_max_f64 %3, %3, %8\n v_max_f64 %4, %4, %8\n v_max_f64 %5, %5, %8\n v_max_f64 %6, %6, %8\n v_max_f64 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if (mode == 7) {   // independent dpp movs
        for (int i = 0; i < REP; i++)
            asm volatile("v_mov_b32_dpp %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(lane));
    } else if (mode == 8) {   // v_cmp_eq_f64 -> sgpr pair, then s_lshr_b64 of it (round trip)
        unsigned long long m = 0;
        for (int i = 0; i < REP; i++)
            asm volatile("v_cmp_eq_f64 vcc, %1, %2\n s_lshr_b64 %0, vcc, 1\n v_cmp_eq_f64 vcc, %1, %2\n s_lshr_b64 %0, vcc, 1\n v_cmp_eq_f64 vcc, %1, %2\n s_lshr_b64 %0, vcc, 1\n v_cmp_eq_f64 vcc, %1, %2\n s_lshr_b64 %0, vcc, 1\n"
                         "v_cmp_eq_f64 vcc, %1, %2\n s_lshr_b64 %0, vcc, 1\n v_cmp_eq_f64 vcc, %1, %2\n s_lshr_b64 %0, vcc, 1\n v_cmp_eq_f64 vcc, %1, %2\n s_lshr_b64 %0, vcc, 1\n v_cmp_eq_f64 vcc, %1, %2\n s_lshr_b64 %0, vcc, 1"
                         : "+s"(m) : "v"(a0), "v"(b) : "vcc", "scc");
        x0 += (int)m;
    } else if (mode == 9) {   // independent v_cmp_eq_f64
        for (int i = 0; i < REP; i++)
            asm volatile("v_cmp_eq_f64 vcc, %0, %1\n v_cmp_eq_f64 vcc, %0, %1\n v_cmp_eq_f64 vcc, %0, %1\n v_cmp_eq_f64 vcc, %0, %1\n v_cmp_eq_f64 vcc, %0, %1\n v_cmp_eq_f64 vcc, %0, %1\n v_cmp_eq_f64 vcc, %0, %1\n v_cmp_eq_f64 vcc, %0, %1"
                         :: "v"(a0), "v"(b) : "vcc");
    } else if (mode == 10) {  // dependent v_add_f64 chain
        for (int i = 0; i < REP; i++)
            asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1"
                         : "+v"(a0) : "v"(b));
    } else if (mode == 11) {  // SALU write -> VALU read -> (v_readfirstlane) -> SALU ... round trip
        for (int i = 0; i < REP; i++)
            asm volatile("v_add_u32 %1, %0, 3\n v_readfirstlane_b32 %0, %1\n v_add_u32 %1, %0, 3\n v_readfirstlane_b32 %0, %1\n v_add_u32 %1, %0, 3\n v_readfirstlane_b32 %0, %1\n v_add_u32 %1, %0, 3\n v_readfirstlane_b32 %0, %1"
                         : "+s"(x0), "+v"(v0));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b + (double)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7) + (double)(v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7);
    if (lane == 0) t[mode] = t1 - t0;
}
int main(){
    double* d; unsigned long long* t; int* idx;
    hipMalloc(&d, 1024*8); hipMalloc(&t, 64*8); hipMalloc(&idx, 64);
    hipMemset(d, 0, 1024*8); hipMemset(t, 0, 64*8); hipMemset(idx, 0, 64);
    const char* nm[] = {"8 indep v_add_f64", "8 indep s_add_u32", "8 indep v_add_u32", "4x (v_add_f64 + s_add) indep", "8 dep s_add_u32", "8 dep v_add_u32", "8 indep v_max_f64", "8 indep v_mov_dpp", "8x (v_cmp->vcc, s_lshr_b64 vcc)", "8 indep v_cmp_eq_f64", "8 dep v_add_f64", "4x (v_add(s) , v_readfirstlane) chain"};
    for (int rep = 0; rep < 2; rep++) for (int m = 0; m < 12; m++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t, idx, m); hipDeviceSynchronize(); }
    unsigned long long ht[64]; hipMemcpy(ht, t, 64*8, hipMemcpyDeviceToHost);
    for (int m = 0; m < 12; m++) printf("%-42s %.1f cycles per group of 8\n", nm[m], (double)ht[m]/REP);
    return 0;
}
