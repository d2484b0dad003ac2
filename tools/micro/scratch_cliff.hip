// Micro-benchmark: cost of per-lane private (scratch) arrays of different sizes on gfx950.
// Usage: ./scratch_cliff   (prints ms per launch of 1M lanes for each scratch size)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int N> __global__ void __launch_bounds__(256) k(int *out, int n, int seed)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int a[N];
    for (int k = 0; k < N; k++) a[k] = k * seed + i;
    int idx = (i * 7 + seed) % N, acc = 0;
    for (int it = 0; it < 64; it++) { acc += a[idx]; idx = (idx * 5 + acc) % N; if (idx < 0) idx = -idx; a[idx] = acc; }
    out[i] = acc;
}
template <int N> void run(int *d, int n)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k<N>, dim3((n + 255) / 256), dim3(256), 0, 0, d, n, 3);
    hipEventRecord(e0, 0);
    for (int w = 0; w < 5; w++) hipLaunchKernelGGL(k<N>, dim3((n + 255) / 256), dim3(256), 0, 0, d, n, 3 + w);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("scratch %6d B/lane: %.3f ms per launch\n", N * 4, ms / 5);
}
int main()
{
    int n = 1 << 20, *d; hipMalloc(&d, n * 4);
    run<16>(d, n); run<64>(d, n); run<128>(d, n); run<256>(d, n); run<512>(d, n); run<1024>(d, n); run<2048>(d, n); run<4096>(d, n);
    return 0;
}
