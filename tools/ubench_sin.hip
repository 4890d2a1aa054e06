// accuracy of sin variants on gfx950 vs a double-precision reference (design input for siren.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
__device__ float sin_hw(float x) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * 0.15915494309189535f)); }
__device__ float sin_hw2(float x) {   // reduce in radians first (2-term Cody-Waite by 2*pi), then hardware sin of the small remainder
    const float j = rintf(x * 0.15915494309189535f);
    float r = fmaf(j, -6.2831854820251465f, x);
    r = fmaf(j, 1.7484555e-7f, r);
    return __builtin_amdgcn_sinf(r * 0.15915494309189535f);
}
__device__ float sin_p9(float x) {    // reduce by pi, odd degree-9 polynomial on [-pi/2, pi/2]
    const float j = rintf(x * 0.3183098861837907f);
    float r = fmaf(j, -3.1415927410125732f, x);
    r = fmaf(j, 8.742278e-8f, r);
    const float z = r * r;
    float p = fmaf(z, 2.7525562e-6f, -1.9840874e-4f);
    p = fmaf(z, p, 8.3333310e-3f);
    p = fmaf(z, p, -1.6666667e-1f);
    p = fmaf(z * r, p, r);
    return ((int)j & 1) ? -p : p;
}
__global__ void k(const float* x, float* y, int n, int mode) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    y[i] = mode == 0 ? sin_hw(x[i]) : mode == 1 ? sin_hw2(x[i]) : mode == 2 ? sin_p9(x[i]) : sinf(x[i]);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> hx(n), hy(n);
    float *dx, *dy; hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4);
    for (float range : {4.0f, 30.0f, 300.0f}) {
        for (int i = 0; i < n; ++i) hx[i] = range * (2.0f * i / n - 1.0f);
        hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 4; ++mode) {
            k<<<n / 256, 256>>>(dx, dy, n, mode);
            hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost);
            double mx = 0; for (int i = 0; i < n; ++i) { double e = fabs((double)hy[i] - sin((double)hx[i])); if (e > mx) mx = e; }
            printf("range %6.0f mode %d (%s): max abs err %.3e\n", range, mode, mode == 0 ? "v_sin(fract(x/2pi))" : mode == 1 ? "CW + v_sin" : mode == 2 ? "pi-reduce + deg9" : "ocml sinf", mx);
        }
    }
    return 0;
}
