// does a lone wave overlap an independent dependent-SALU chain with a dependent-VALU chain?
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 128
__global__ void k(double* out, unsigned long long* t, int mode, int* idx)
{
    double a = out[threadIdx.x], b = out[threadIdx.x+64];
    unsigned s = __builtin_amdgcn_readfirstlane(idx[0]);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {        // SALU chain only
        #pragma unroll
        for (int i = 0; i < REP; i++) asm volatile("s_mul_i32 %0, %1, 0xc9" : "=s"(s) : "s"(s) : "scc");
    } else if (mode == 1) { // VALU chain only
        #pragma unroll
        for (int i = 0; i < REP; i++) asm volatile("v_add_f64 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
    } else if (mode == 2) { // interleaved 1:1
        #pragma unroll
        for (int i = 0; i < REP; i++) { asm volatile("s_mul_i32 %0, %1, 0xc9" : "=s"(s) : "s"(s) : "scc"); asm volatile("v_add_f64 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b)); }
    } else if (mode == 3) { // two independent VALU chains interleaved
        double c = b * 0.5;
        #pragma unroll
        for (int i = 0; i < REP; i++) { asm volatile("v_add_f64 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b)); asm volatile("v_add_f64 %0, %1, %2" : "=v"(c) : "v"(c), "v"(b)); }
        a += c;
    } else if (mode == 4) { // VALU -> SGPR -> SALU -> VALU round trip: v_cmp, s_ff1, v_add with sgpr
        unsigned long long m;
        #pragma unroll
        for (int i = 0; i < REP; i++) { asm volatile("v_cmp_eq_f64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(a)); asm volatile("s_ff1_i32_b64 %0, %1" : "=s"(s) : "s"(m) : "scc"); asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a) : "s"(s)); }
    } else if (mode == 5) { // v_cmp -> sgpr -> s_lshr_b64 -> s_ff1 -> s_mul -> v_add_u32 (address chain)
        unsigned long long m; unsigned v = threadIdx.x;
        #pragma unroll
        for (int i = 0; i < REP; i++) { asm volatile("v_cmp_eq_u32 %0, %1, %2" : "=s"(m) : "v"(v), "v"(v)); asm volatile("s_lshr_b64 %0, %1, %2" : "=s"(m) : "s"(m), "s"(s) : "scc"); asm volatile("s_ff1_i32_b64 %0, %1" : "=s"(s) : "s"(m) : "scc"); asm volatile("s_mul_i32 %0, %1, 0xc8" : "=s"(s) : "s"(s) : "scc"); asm volatile("v_add_u32 %0, %1, %2" : "=v"(v) : "s"(s), "v"(v)); }
        a += v;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a + (double)s;
    if (threadIdx.x == 0) t[mode] = t1 - t0;
}
int main(){
    double* d; unsigned long long* t; int* idx; hipMalloc(&d, 1024*8); hipMalloc(&t, 64*8); hipMalloc(&idx, 64); hipMemset(d,0,1024*8); hipMemset(t,0,64*8); hipMemset(idx,0,64);
    const char* nm[]={"dep SALU (s_mul) chain","dep VALU (v_add_f64) chain","interleaved SALU+VALU chains","two interleaved VALU chains","v_cmp->sgpr->s_ff1->v_cvt round trip","v_cmp->s_lshr->s_ff1->s_mul->v_add_u32"};
    for (int r=0;r<2;r++) for (int m=0;m<6;m++){ hipLaunchKernelGGL(k,dim3(1),dim3(64),0,0,d,t,m,idx); hipDeviceSynchronize(); }
    unsigned long long ht[64]; hipMemcpy(ht,t,64*8,hipMemcpyDeviceToHost);
    for (int m=0;m<6;m++) printf("%-45s %.1f cycles per iteration\n", nm[m], (double)ht[m]/REP);
    return 0;
}
