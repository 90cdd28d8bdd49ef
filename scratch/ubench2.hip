// timing-only variants of the speculative walker body (LC=5, NODEL), one wave, fake table in LDS
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define LC 5
#define ROW 25
#define BLK 150
#define NSRC 40
template <int CTRL> __device__ __forceinline__ double dppf(double v){int lo=__builtin_amdgcn_mov_dpp(__double2loint(v),CTRL,0xF,0xF,true);int hi=__builtin_amdgcn_mov_dpp(__double2hiint(v),CTRL,0xF,0xF,true);return __hiloint2double(hi,lo);}
__device__ __forceinline__ double vmax(double a,double b){double r;asm("v_max_f64 %0, %1, %2":"=v"(r):"v"(a),"v"(b));return r;}
__device__ __forceinline__ unsigned long long amax(double acc){double m=acc;m=vmax(m,dppf<0xB1>(m));m=vmax(m,dppf<0x4E>(m));return __builtin_amdgcn_ballot_w64(acc==m);}

// MODE bits: 1 = real R (w-dependent row reads), 2 = real M (dpp argmax), 4 = real S (adds), 8 = sched_barrier at end, 16 = sched_barrier between A|S and R|M
template <int MODE>
__global__ void k(double* out, unsigned long long* t, int slot, int iters)
{
    __shared__ double g[(NSRC+2)*BLK];
    for (int i = threadIdx.x; i < (NSRC+2)*BLK; i += 64) g[i] = -(double)((i*7919)%1013) * 0.001 - (double)(i%5==(i/150)%4)*0.5;
    __syncthreads();
    const int lane = threadIdx.x, bb = lane & 3, ga = (lane>>3) < 6 ? (lane>>3) : 5;
    double Y[LC][LC];
    #pragma unroll
    for (int u=0;u<LC;u++)
    #pragma unroll
    for (int l=0;l<LC;l++) Y[u][l]=0.0;
    int sh = 40; unsigned long long B = amax(g[bb+ga*ROW]); double hyp = g[bb+BLK+ga*ROW];
    unsigned long long word = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        const double* gb = g + bb; const double* gh = gb + ga*ROW;
        for (int gq = 0; gq < NSRC/LC; gq++) {
            #pragma unroll
            for (int u = 0; u < LC; u++) {
                const int s = gq*LC+u;
                const int w = (int)__builtin_ctzll(B >> sh) & 3;
                sh = 8*w; word = (word<<4)|w;
                double acc = hyp;
                if (MODE & 4) {
                    #pragma unroll
                    for (int l=1;l<LC;l++) acc += Y[(u-(l-1)+LC)%LC][l];
                } else acc += Y[u][1];
                if (MODE & 16) __builtin_amdgcn_sched_barrier(0);
                const double* row = gb + (s+1)*BLK + ((MODE&1)? w*ROW : 0);
                #pragma unroll
                for (int l=1;l<LC;l++) Y[(u+1)%LC][l] = row[l*5];
                hyp = gh[(s+2)*BLK];
                if (MODE & 2) B = amax(acc); else B = __builtin_amdgcn_ballot_w64(acc < -1.0) | 0x0101010101010101ull;
                if (MODE & 8) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = hyp + (double)word + Y[0][1];
    if (threadIdx.x == 0) t[slot] = t1 - t0;
}
int main(){
    double* d; unsigned long long* t; hipMalloc(&d, 1024*8); hipMalloc(&t, 64*8); hipMemset(t,0,64*8);
    const int iters = 50; const double steps = iters*NSRC;
    #define RUN(M,slot) for(int r=0;r<2;r++){hipLaunchKernelGGL(k<M>,dim3(1),dim3(64),0,0,d,t,slot,iters);hipDeviceSynchronize();}
    RUN(7,0) RUN(15,1) RUN(31,2) RUN(6,3) RUN(5,4) RUN(3,5) RUN(4,6) RUN(0,7) RUN(23,8)
    unsigned long long ht[64]; hipMemcpy(ht,t,64*8,hipMemcpyDeviceToHost);
    const char* nm[]={"full (R+M+S)","full + barrier at end","full + both barriers","no R dependence (M+S)","no M (R+S)","no S chain (R+M)","S only","A only (+1 add)","full + mid barrier only"};
    for(int i=0;i<9;i++) printf("%-28s %.1f cycles/step\n", nm[i], ht[i]/steps);
    return 0;
}
