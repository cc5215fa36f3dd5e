// micro-benchmark: streaming 48-byte records (3 x b128 per lane, stride 48) vs a linear b128 stream of the same bytes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Rec { double2 a, b, c; };
__global__ void __launch_bounds__(512) k_aos(const Rec *in, Rec *out, long n)
{
    for (long i = blockIdx.x * 512L + threadIdx.x; i < n; i += gridDim.x * 512L) {
        Rec r = in[i];
        r.a.x += 1.0; r.b.x += 1.0; r.c.x += 1.0;
        out[i] = r;
    }
}
__global__ void __launch_bounds__(512) k_lin(const double2 *in, double2 *out, long n16)
{
    for (long i = blockIdx.x * 512L + threadIdx.x; i < n16; i += gridDim.x * 512L) {
        double2 v = in[i]; v.x += 1.0; out[i] = v;
    }
}
// per-wave linear: wave reads its 64 records as 3 fully coalesced 1 KB pieces
__global__ void __launch_bounds__(512) k_wavelin(const double2 *in, double2 *out, long n)
{
    const int lane = threadIdx.x & 63;
    for (long w = (blockIdx.x * 512L + threadIdx.x) >> 6; w * 64 < n; w += (gridDim.x * 512L) >> 6) {
        const long base = w * 192; // 64 records * 3 double2
        double2 v0 = in[base + lane], v1 = in[base + 64 + lane], v2 = in[base + 128 + lane];
        v0.x += 1.0; v1.x += 1.0; v2.x += 1.0;
        out[base + lane] = v0; out[base + 64 + lane] = v1; out[base + 128 + lane] = v2;
    }
}
int main()
{
    const long n = 502584 * 8L; // 8x the 1M-mesh node count: 193 MB in, 193 MB out (beyond L2; partly MALL)
    Rec *a, *b; hipMalloc(&a, n * 48); hipMalloc(&b, n * 48); hipMemset(a, 0, n * 48);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {512, 1024, 2048}) for (int which = 0; which < 3; ++which) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            for (int it = 0; it < 10; ++it) {
                if (which == 0) k_aos<<<grid, 512>>>(a, b, n);
                else if (which == 1) k_lin<<<grid, 512>>>((double2 *)a, (double2 *)b, n * 3);
                else k_wavelin<<<grid, 512>>>((double2 *)a, (double2 *)b, n);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("grid %4d %-8s %.1f us/launch  %.2f TB/s (read+write)\n", grid, which == 0 ? "aos" : which == 1 ? "linear" : "wavelin",
               best * 100, 2.0 * n * 48 / (best / 10 * 1e-3) / 1e12);
    }
    return 0;
}
