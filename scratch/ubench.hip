// micro-benchmarks for the walker's instruction mix: one wave, s_memtime around N repeats
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 256
__device__ __forceinline__ double dpp_f64_b1(double v){int lo=__builtin_amdgcn_mov_dpp(__double2loint(v),0xB1,0xF,0xF,true);int hi=__builtin_amdgcn_mov_dpp(__double2hiint(v),0xB1,0xF,0xF,true);return __hiloint2double(hi,lo);}
__device__ __forceinline__ double vmax(double a,double b){double r;asm volatile("v_max_f64 %0, %1, %2":"=v"(r):"v"(a),"v"(b));return r;}
__device__ __forceinline__ double vadd(double a,double b){double r;asm volatile("v_add_f64 %0, %1, %2":"=v"(r):"v"(a),"v"(b));return r;}

__global__ void k(double* out, unsigned long long* t, int mode, int* idx)
{
    __shared__ double lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (double)((i*7)%13);
    __syncthreads();
    double a = out[threadIdx.x], b = out[threadIdx.x+64], c = a*0.5, d=b*0.5;
    int w = idx[0];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) { // dependent v_add_f64 chain
        #pragma unroll
        for (int i = 0; i < REP; i++) a = vadd(a, b);
    } else if (mode == 1) { // 4 independent chains
        #pragma unroll
        for (int i = 0; i < REP/4; i++) { a = vadd(a,b); c = vadd(c,b); d = vadd(d,b); b = vadd(b, 1.0);}    
    } else if (mode == 2) { // dpp + max chain (one argmax level per iter)
        #pragma unroll
        for (int i = 0; i < REP; i++) a = vmax(a, dpp_f64_b1(a));
    } else if (mode == 3) { // dependent LDS read chain (address from loaded value)
        #pragma unroll
        for (int i = 0; i < REP; i++) { double v = lds[(w & 1023) + (threadIdx.x&7)]; w = (int)v + w; }
        a += w;
    } else if (mode == 4) { // SALU chain: shift / ff1 / and / mul
        unsigned long long B = 0x0101010101010101ull * (unsigned)w;
        #pragma unroll
        for (int i = 0; i < REP; i++) { int x = (int)(__builtin_ctzll((B >> (8 * (w&7))) | 256) & 7); w = __builtin_amdgcn_readfirstlane(x * 200 + w); }
        a += w;
    } else if (mode == 5) { // v_cmp -> ballot -> ff1 -> scalar mul -> v_add address -> ds_read -> v_add_f64 (one non-spec step skeleton)
        #pragma unroll
        for (int i = 0; i < REP; i++) {
            unsigned long long m = __builtin_amdgcn_ballot_w64(a == c);
            int x = __builtin_amdgcn_readfirstlane((int)(__builtin_ctzll(m | 0x100) & 7));
            double v = lds[x * 25 + (threadIdx.x & 7) + (i&7)*200];
            a = vadd(v, b); c = vmax(a, dpp_f64_b1(a));
        }
    } else if (mode == 6) { // independent v_mov dpp throughput
        int x = (int)a, y = (int)b;
        #pragma unroll
        for (int i = 0; i < REP/2; i++) { x = __builtin_amdgcn_mov_dpp(x,0xB1,0xF,0xF,true) + 1; y = __builtin_amdgcn_mov_dpp(y,0x4E,0xF,0xF,true)+1; }
        a += x + y;
    } else if (mode == 7) { // empty
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a + c + d + b;
    if (threadIdx.x == 0) t[mode] = t1 - t0;
}
int main(){
    double* d; unsigned long long* t; int* idx;
    hipMalloc(&d, 1024*8); hipMalloc(&t, 64*8); hipMalloc(&idx, 64);
    hipMemset(d, 0, 1024*8); hipMemset(t, 0, 64*8); hipMemset(idx, 0, 64);
    const char* names[] = {"dep v_add_f64", "4 indep v_add_f64 chains", "dpp(2)+v_max_f64 dep", "dep LDS read chain (+cvt,+add)", "SALU shift/ff1/and/mul/add chain", "step skeleton cmp->ff1->lds->add->dpp/max", "v_mov_dpp+add 2 chains", "empty"};
    for (int rep = 0; rep < 2; rep++) for (int m = 0; m < 8; m++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t, m, idx); hipDeviceSynchronize(); }
    unsigned long long ht[64]; hipMemcpy(ht, t, 64*8, hipMemcpyDeviceToHost);
    for (int m = 0; m < 8; m++) printf("%-45s %8llu cycles  %.1f per iter\n", names[m], ht[m], (double)(ht[m]-ht[7])/REP);
    return 0;
}
