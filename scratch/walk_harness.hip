// standalone harness: run k_walk_spec<LC> at speculation depth 1 (variant 0) and 2 (variant 1) on a random table,
// compare paths, print cycles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "../include/gretel_hip.h"
#include "../include/gh_detlog.h"
#include "../gretel_amd/csrc/kernels.hpp"
#ifndef HLC
#define HLC 5
#endif
int main(int argc, char** argv){
    const int N = argc > 1 ? atoi(argv[1]) : 2000, LC = HLC;
    const int variant_only = argc > 2 ? atoi(argv[2]) : -1;
    const size_t nG = (size_t)(N + LT_PAD) * 6 * LC * 5;
    std::vector<double> G(nG), minfo((size_t)(N + 2) * MINFO, 0.0);
    srand(1);
    for (size_t i = 0; i < nG; i++) G[i] = -(double)(rand() % 100000) * 1e-4;
    for (int p = 0; p <= N + 1; p++) { for (int q = 0; q < 16; q++) minfo[(size_t)p*MINFO+q] = -0.3; for (int q=5;q<10;q++) minfo[(size_t)p*MINFO+q]=0.25; long long cm=15; memcpy(&minfo[(size_t)p*MINFO+10], &cm, 8); }
    // the depth-2 walker's tables, as k_lt derives them from G (kernels.hpp): Ht = x1 + x2, Yt = the resolved lags
    const int nyp = deep_nyp(LC), ypos = 16 * nyp, nsrc_all = N + LT_PAD;
    std::vector<double> Ht((size_t)(N + WALK_TPAD) * 64, 0.0), Yt((size_t)(N + WALK_TPAD) * (ypos ? ypos : 1), 0.0);
    auto Gat = [&](int i, int row, int lag, int col) { return G[(((size_t)i * 6 + row) * LC + (lag - 1)) * 5 + col]; };
    for (int tt = 1; tt < N + WALK_TPAD; tt++)
        for (int ln = 0; ln < 64; ln++) {
            const int b = ln & 3, a1 = (ln >> 2) & 3, a2 = ln >> 4;
            double v = 0.0;
            if (tt - 1 < nsrc_all) {
                v = Gat(tt - 1, tt - 1 == 0 ? 5 : a1, 1, b);
                if (tt >= 2 && LC >= 2) v = v + Gat(tt - 2, tt - 2 == 0 ? 5 : a2, 2, b);
            }
            Ht[(size_t)tt * 64 + ln] = v;
        }
    for (int i = 0; i < N + WALK_TPAD && ypos; i++)
        for (int r = 0; r < ypos; r++) {
            const int wb = r / nyp, li = r % nyp;
            if (i < nsrc_all && li + 2 < LC) Yt[(size_t)i * ypos + r] = Gat(i, i == 0 ? 5 : (wb >> 2), li + 3, wb & 3);
        }
    double *dHt, *dYt;
    hipMalloc(&dHt, Ht.size() * 8); hipMalloc(&dYt, Yt.size() * 8);
    hipMemcpy(dHt, Ht.data(), Ht.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dYt, Yt.data(), Yt.size() * 8, hipMemcpyHostToDevice);
    const int tables = argc > 3 ? atoi(argv[3]) : 1;      // 0: the depth-2 loaders gather from G instead of copying the tables
    double *dG, *dmi; uint8_t* dpath[2]; gh_path_rec* drec; dev_state* dst;
    hipMalloc(&dG, nG*8); hipMalloc(&dmi, minfo.size()*8); hipMalloc(&dpath[0], N+2); hipMalloc(&dpath[1], N+2); hipMalloc(&drec, sizeof(gh_path_rec)); hipMalloc(&dst, sizeof(dev_state));
    hipMemcpy(dG, G.data(), nG*8, hipMemcpyHostToDevice); hipMemcpy(dmi, minfo.data(), minfo.size()*8, hipMemcpyHostToDevice);
    const int chunk = walk_chunk(LC, false);
    const size_t lds = walk_lds_bytes(LC, false) > walk_lds_bytes(LC, true) ? walk_lds_bytes(LC, false) : walk_lds_bytes(LC, true);
    hipFuncSetAttribute((const void*)k_walk_spec<HLC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<uint8_t> path[2]; 
    for (int v = 0; v < 2; v++) {
        path[v].assign(N+1, 255);
        if (variant_only >= 0 && v != variant_only) continue;
        dev_state hs;
        double best = 1e30;
        hipError_t e = hipSuccess;
        for (int rep = 0; rep < 4; rep++) {         // first launch is cold: report the best of four
            memset(&hs, 0, sizeof hs); hs.first_hole = 0x7f7f7f7f; hs.nodel = 1; hs.narrow = 1; hs.ranked = v;   // ranked tables <=> depth-2 walker
            hipMemcpy(dst, &hs, sizeof hs, hipMemcpyHostToDevice);
            walk_params P; P.N = N; P.L = LC; P.chunk = chunk; P.rearm = 0; P.depth2 = v; P.G = dG; P.Ht = tables ? dHt : nullptr; P.Yt = tables ? dYt : nullptr; P.minfo = dmi; P.path_out = dpath[v]; P.rec = drec; P.st = dst; P.min_remove = 0.01;
            hipLaunchKernelGGL((k_walk_spec<HLC>), dim3(1), dim3(512), lds, 0, P, (const win_desc*)nullptr, 0);
            e = hipDeviceSynchronize();
            hipMemcpy(&hs, dst, sizeof hs, hipMemcpyDeviceToHost);
            if (hs.dbg[2] && (double)hs.dbg[0]/hs.dbg[2] < best) best = (double)hs.dbg[0]/hs.dbg[2];
        }
        hipMemcpy(path[v].data(), dpath[v], N+1, hipMemcpyDeviceToHost);
        printf("variant %d: %s  cycles/step %.1f  n_done %d\n", v, hipGetErrorString(e), best, hs.n_done); fflush(stdout);
#ifdef GH_STAMPS
        { const char* nm[5] = {"loop/M-tail->A", "A resolve+pack", "S adds", "R reads issue", "M argmax"}; for (int q = 0; q < 5; q++) printf("   seg %d %-16s %.1f cycles/step (incl. ~stamp cost)\n", q, nm[q], (double)hs.dbg8[q]/hs.dbg[2]); }
#endif
    }
    if (variant_only < 0) { int diff = 0; for (int i = 0; i <= N; i++) diff += path[0][i] != path[1][i]; printf("path differences: %d of %d\n", diff, N+1); }
    return 0;
}
