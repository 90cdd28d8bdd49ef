// column vs row reads of the band block of a position, float vs double (k_rw's round-2 access pattern at C3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <typename T, bool COL, bool WIDE>
__global__ void __launch_bounds__(256) k(const T *band, int N, int W, const unsigned char *path, double *out)
{
    const int t = blockIdx.x * 256 + threadIdx.x, p = t >> 3, s = t & 7;
    if (p > N - 8 || s >= W) return;
    const int a = path[p], b = path[p + s + 1];
    const T *base = band + ((size_t)p * 7 * W * 7);
    T v[7];
    if (WIDE) {
        // the 16-byte chunk that holds the element, then pick
        for (int x = 0; x < 7; x++) {
            const size_t e = COL ? ((size_t)x * W + s) * 7 + b : ((size_t)a * W + s) * 7 + x;
            const size_t e4 = e & ~(size_t)(16 / sizeof(T) - 1);
            typedef T vec __attribute__((ext_vector_type(16 / sizeof(T))));
            const vec q = *reinterpret_cast<const vec *>(base + e4);
            T r = q[0];
            for (unsigned k2 = 1; k2 < 16 / sizeof(T); k2++) r = (e - e4 == k2) ? q[k2] : r;
            v[x] = r;
        }
    } else {
#pragma unroll
        for (int x = 0; x < 7; x++) v[x] = COL ? base[((size_t)x * W + s) * 7 + b] : base[((size_t)a * W + s) * 7 + x];
    }
    double acc = 0;
#pragma unroll
    for (int x = 0; x < 7; x++) acc += (double)v[x];
    if (acc == 12345.678) out[0] = acc;
}
template <typename T, bool COL, bool WIDE> void run(const char *name, int N, int W)
{
    T *band; unsigned char *path; double *out;
    const size_t n = (size_t)(N + 2) * 7 * W * 7;
    hipMalloc(&band, n * sizeof(T)); hipMalloc(&path, N + 64); hipMalloc(&out, 8);
    std::vector<T> hb(n); for (size_t i = 0; i < n; i++) hb[i] = (T)(i % 13);
    std::vector<unsigned char> hp(N + 64); for (int i = 0; i < N + 64; i++) hp[i] = (i * 7 + i / 3) % 4;
    hipMemcpy(band, hb.data(), n * sizeof(T), hipMemcpyHostToDevice); hipMemcpy(path, hp.data(), N + 64, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nb = ((N + 1) * 8 + 255) / 256;
    float best = 1e9;
    for (int it = 0; it < 20; it++) {
        hipEventRecord(e0); hipLaunchKernelGGL((k<T, COL, WIDE>), dim3(nb), dim3(256), 0, 0, band, N, W, path, out); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-28s N=%d W=%d: %.1f us\n", name, N, W, best * 1e3);
    hipFree(band); hipFree(path); hipFree(out);
}
int main()
{
    for (int W : {4, 20}) {
        const int N = W == 4 ? 10000 : 50000;
        run<float, false, false>("float row", N, W); run<float, true, false>("float col", N, W); run<float, true, true>("float col 16B", N, W);
        run<double, false, false>("double row", N, W); run<double, true, false>("double col", N, W); run<double, true, true>("double col 16B", N, W);
    }
    return 0;
}
