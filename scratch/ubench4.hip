// fully pinned (asm volatile) skeleton of the speculative walker body, LC=5 NODEL, to find the best order.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
#define ROWB 200
#define BLKB 1200
#define NSRC 40
#define DPPMOV(dlo,dhi,s,ctrl) { int lo_=__builtin_amdgcn_mov_dpp(__double2loint(s),ctrl,0xF,0xF,true); int hi_=__builtin_amdgcn_mov_dpp(__double2hiint(s),ctrl,0xF,0xF,true); t_=__hiloint2double(hi_,lo_);} 
#define VADD(d,a,b) asm volatile("v_add_f64 %0, %1, %2":"=v"(d):"v"(a),"v"(b))
#define VMAX(d,a,b) asm volatile("v_max_f64 %0, %1, %2":"=v"(d):"v"(a),"v"(b))
#define A1() asm volatile("s_lshr_b64 %0, %1, %2":"=s"(q):"s"(B),"s"(sh):"scc")
#define A2() asm volatile("s_ff1_i32_b64 %0, %1":"=s"(w):"s"(q):"scc")
#define A3() asm volatile("s_mul_i32 %0, %1, 0xc8":"=s"(off):"s"(w):"scc")
#define A4() asm volatile("s_lshl_b32 %0, %1, 3":"=s"(sh):"s"(w):"scc")
#define A5() asm volatile("v_add_u32 %0, %1, %2":"=v"(vrow):"s"(off),"v"(vbase))
#define PK() asm volatile("s_lshl4_add_u32 %0, %1, %2":"=s"(word):"s"(word),"s"(w):"scc")
#define RD() { asm volatile("ds_read2_b64 %0, %1 offset0:5 offset1:10":"=v"(ya):"v"(vrow)); asm volatile("ds_read2_b64 %0, %1 offset0:15 offset1:20":"=v"(yb):"v"(vrow)); asm volatile("ds_read_b64 %0, %1 offset:2400":"=v"(hyp):"v"(vhyp)); }
#define WAIT() asm volatile("s_waitcnt lgkmcnt(0)":::"memory")
#define CMP() asm volatile("v_cmp_eq_f64 %0, %1, %2":"=s"(B):"v"(acc),"v"(m))

template <int MODE>
__global__ void k(double* out, unsigned long long* t, int slot, int iters)
{
    __shared__ double g[(NSRC+4)*150];
    for (int i = threadIdx.x; i < (NSRC+4)*150; i += 64) g[i] = -(double)((i*7919)%1013) * 0.001;
    __syncthreads();
    const int lane = threadIdx.x, bb = lane & 3, ga = (lane>>3) < 6 ? (lane>>3) : 5;
    unsigned vbase0 = (unsigned)(unsigned long long)g + bb*8, vbase = vbase0, vhyp = vbase0 + ga*ROWB, vrow = vbase0;
    unsigned sh = 0, w = 0, off = 0, word = 0; unsigned long long B = 0x0101010101010101ull, q;
    d2 ya = {-1.0,-2.0}, yb = {-3.0,-4.0}, ya1 = ya, yb1 = yb, ya2 = ya, yb2 = yb; double hyp = -0.5, acc = -1.0, m, t_;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        vbase = vbase0; vhyp = vbase0 + ga*ROWB;
        for (int s = 0; s < NSRC; s++) {
            // rotate (cheap approximation of the register rotation: older rows)
            if (MODE == 0) {            // serial: A R S M
                A1(); A2(); A3(); A4(); A5(); PK();
                d2 pa = ya, pb = yb; double ph = hyp;
                RD();
                WAIT();
                VADD(acc, ph, pa.x); VADD(acc, acc, ya1.y); VADD(acc, acc, yb2.x); VADD(acc, acc, yb1.y);
                ya2 = ya1; yb2 = yb1; ya1 = pa; yb1 = pb;
                DPPMOV(0,0,acc,0xB1); VMAX(m, acc, t_); DPPMOV(0,0,m,0x4E); VMAX(m, m, t_); CMP();
            } else if (MODE == 1) {     // A || S interleaved, then R, then M   (reads waited at top of next body)
                WAIT();
                A1(); VADD(acc, hyp, ya.x); A2(); VADD(acc, acc, ya1.y); A3(); VADD(acc, acc, yb2.x); A4(); A5(); VADD(acc, acc, yb1.y); PK();
                ya2 = ya1; yb2 = yb1; ya1 = ya; yb1 = yb;
                RD();
                DPPMOV(0,0,acc,0xB1); VMAX(m, acc, t_); DPPMOV(0,0,m,0x4E); VMAX(m, m, t_); CMP();
            } else if (MODE == 2) {     // S first, then M with A's scalar ops in its shadows, R last
                WAIT();
                VADD(acc, hyp, ya.x); VADD(acc, acc, ya1.y); VADD(acc, acc, yb2.x); VADD(acc, acc, yb1.y);
                ya2 = ya1; yb2 = yb1; ya1 = ya; yb1 = yb;
                A1(); DPPMOV(0,0,acc,0xB1); A2(); VMAX(m, acc, t_); A3(); A4(); DPPMOV(0,0,m,0x4E); A5(); VMAX(m, m, t_); PK();
                RD();
                CMP();
            } else if (MODE == 3) {     // A first (needs previous ballot), R immediately, then S and M
                A1(); A2(); A3(); A4(); A5(); PK();
                d2 pa = ya, pb = yb; double ph = hyp;
                RD();
                VADD(acc, ph, pa.x); VADD(acc, acc, ya1.y); VADD(acc, acc, yb2.x); VADD(acc, acc, yb1.y);
                ya2 = ya1; yb2 = yb1; ya1 = pa; yb1 = pb;
                DPPMOV(0,0,acc,0xB1); VMAX(m, acc, t_); DPPMOV(0,0,m,0x4E); VMAX(m, m, t_);
                WAIT();
                CMP();
            }
            vbase += BLKB; vhyp += BLKB;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = hyp + (double)word + ya.x + yb.y + acc;
    if (threadIdx.x == 0) t[slot] = t1 - t0;
}
int main(){
    double* d; unsigned long long* t; hipMalloc(&d, 1024*8); hipMalloc(&t, 64*8); hipMemset(t,0,64*8);
    const int iters = 50; const double steps = iters*NSRC;
    #define RUN(M) for(int r=0;r<2;r++){hipLaunchKernelGGL(k<M>,dim3(1),dim3(64),0,0,d,t,M,iters);hipDeviceSynchronize();}
    RUN(0) RUN(1) RUN(2) RUN(3)
    unsigned long long ht[64]; hipMemcpy(ht,t,64*8,hipMemcpyDeviceToHost);
    const char* nm[]={"serial A R wait S M","wait, A||S, R, M","wait, S, M with A in shadows, R","A R S M, wait before cmp"};
    for(int i=0;i<4;i++) printf("%-40s %.1f cycles/step\n", nm[i], ht[i]/steps);
    return 0;
}
