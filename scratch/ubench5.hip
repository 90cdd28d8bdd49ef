// one-wave throughput of the depth-2 body's pieces: which combination costs what
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 1000
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v){int lo=__builtin_amdgcn_mov_dpp(__double2loint(v),CTRL,0xF,0xF,true);int hi=__builtin_amdgcn_mov_dpp(__double2hiint(v),CTRL,0xF,0xF,true);return __hiloint2double(hi,lo);}
__device__ __forceinline__ double vmax(double a,double b){double r;asm("v_max_f64 %0, %1, %2":"=v"(r):"v"(a),"v"(b));return r;}
typedef __attribute__((address_space(3))) const double lds_cdouble;

template <int MODE>
__global__ void k(double* out, unsigned long long* t, const int* idx)
{
    __shared__ double lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = -(double)((i*7)%13) - 1.0;
    __syncthreads();
    const int lane = threadIdx.x;
    double accP = out[lane], acc2 = out[lane + 64], y0 = -1.0, y1 = -2.0, y2 = -3.0, h1 = -0.5, h2 = -0.25, H12 = -1.0;
    unsigned long long B = 0x1111111111111111ull * (unsigned long long)(idx[0] + 1);
    int sh = idx[1];
    unsigned long long word = 0;
    const unsigned base = (unsigned)(uintptr_t)lds + (lane & 3) * 8;
    unsigned rowb; asm("v_mov_b32 %0, 200" : "=v"(rowb));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REP; i++) {
        int w = 0;
        if (MODE & 1) {          // A
            w = (int)__builtin_ctzll(B >> sh) & 3;
            sh = (w << 4) | ((sh >> 2) & 12);
            word = (word << 4) | (unsigned long long)w;
        }
        if (MODE & 2) {          // M
            double m = vmax(accP, dpp_f64<0xB1>(accP));
            m = vmax(m, dpp_f64<0x4E>(m));
            B = __builtin_amdgcn_ballot_w64(accP == m);
        }
        if (MODE & 4) {          // S
            double acc = H12; acc += y0; acc += y1; acc += y2;
            accP = acc; H12 = h1 + h2;
        } else if (MODE & 2) accP = acc2 + (double)(i & 3);
        if (MODE & 8) {          // R
            unsigned vrow;
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(vrow) : "s"(w), "v"(rowb), "v"(base + (unsigned)(i & 7) * 1200u));
            lds_cdouble* row = (lds_cdouble*)vrow;
            y2 = y1; y1 = y0;   // rotate (costless renames after unrolling in the real kernel; here v_movs)
            y0 = row[10]; double t1 = row[15], t2 = row[20];
            y1 += 0.0 * t1; y2 += 0.0 * t2;
            h1 = *(lds_cdouble*)(base + 4800u + (unsigned)(i & 7) * 1200u);
            h2 = *(lds_cdouble*)(base + 3600u + 40u + (unsigned)(i & 7) * 1200u);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[lane] = accP + H12 + y0 + y1 + y2 + (double)sh + (double)(word & 1023) + (double)(B & 7);
    if (lane == 0) t[MODE] = t1 - t0;
}
#define RUN(m) hipLaunchKernelGGL(k<m>, dim3(1), dim3(64), 0, 0, d, t, idx); hipDeviceSynchronize();
int main(){
    double* d; unsigned long long* t; int* idx;
    hipMalloc(&d, 1024*8); hipMalloc(&t, 64*8); hipMalloc(&idx, 64);
    hipMemset(d, 0, 1024*8); hipMemset(t, 0, 64*8); hipMemset(idx, 0, 64);
    for (int rep = 0; rep < 2; rep++) { RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(6) RUN(7) RUN(8) RUN(9) RUN(11) RUN(15) RUN(14) RUN(5) }
    unsigned long long ht[64]; hipMemcpy(ht, t, 64*8, hipMemcpyDeviceToHost);
    const int modes[] = {0,1,2,3,4,5,6,7,8,9,11,14,15};
    for (int m : modes) printf("mode %2d [%s%s%s%s] %.1f cycles/iter\n", m, m&1?"A":"-", m&2?"M":"-", m&4?"S":"-", m&8?"R":"-", (double)(ht[m]-ht[0])/REP);
    return 0;
}
