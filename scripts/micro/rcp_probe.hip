// How accurate is v_rcp_f64 on gfx950, and how many Newton steps does a full-precision reciprocal need?
// hipcc --offload-arch=gfx950 -O3 scripts/micro/rcp_probe.hip -o /tmp/rcp_probe && /tmp/rcp_probe
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>

__global__ void k(const double *a, double *seed, double *one, double *two, int n)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double y = __builtin_amdgcn_rcp(a[i]);
    seed[i] = y;
    double e = fma(-a[i], y, 1.0);
    y = fma(y, e, y);
    one[i] = y;
    e = fma(-a[i], y, 1.0);
    y = fma(y, e, y);
    two[i] = y;
}

int main()
{
    const int n = 1 << 22;
    double *h = (double *)malloc(8 * n), *r[3];
    srand48(7);
    for (int i = 0; i < n; ++i) h[i] = ldexp(1.0 + drand48(), (int)(lrand48() % 80) - 40) * (lrand48() & 1 ? 1 : -1);
    double *d, *o[3];
    hipMalloc(&d, 8 * n);
    hipMemcpy(d, h, 8 * n, hipMemcpyHostToDevice);
    for (int k_ = 0; k_ < 3; ++k_) {
        hipMalloc(&o[k_], 8 * n);
        r[k_] = (double *)malloc(8 * n);
    }
    k<<<n / 256, 256>>>(d, o[0], o[1], o[2], n);
    for (int k_ = 0; k_ < 3; ++k_) hipMemcpy(r[k_], o[k_], 8 * n, hipMemcpyDeviceToHost);
    const char *name[3] = {"v_rcp_f64 seed", "one Newton step", "two Newton steps"};
    for (int k_ = 0; k_ < 3; ++k_) {
        double worst = 0;
        long exact = 0;
        for (int i = 0; i < n; ++i) {
            const double t = 1.0 / h[i];
            const double rel = fabs((r[k_][i] - t) / t);
            if (rel > worst) worst = rel;
            if (r[k_][i] == t) ++exact;
        }
        printf("%-18s max rel err %.3e (%.1f ulp of 2^-53), correctly rounded in %.4f %% of %d samples\n", name[k_], worst,
               worst / 1.1102230246251565e-16, 100.0 * exact / n, n);
    }
    return 0;
}
