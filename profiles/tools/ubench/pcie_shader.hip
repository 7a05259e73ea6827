// pcie_shader.hip -- does the host link give both directions at once when one of them is driven by a kernel instead of SDMA?
//   hipcc --offload-arch=gfx950 -O3 -o pcie_shader pcie_shader.hip ; ./pcie_shader
// Rates in GB/s for 1 GiB per direction, pinned host memory:
//   sdma   : hipMemcpyAsync on its own stream
//   shader : a kernel of W workgroups x 256 threads moving 16 bytes per lane and trip (host pointer mapped into the device)
// "A | B" = both at once, each on its own stream; the rate of each is total bytes / the time until BOTH are done, and the per-
// direction times are printed too.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // four loads in flight per lane
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
// a line-shaped egress: records of `rec` bytes at UNALIGNED destinations (a SAM line with a hole), byte stores at the edges
__global__ void __launch_bounds__(64) copy_lines(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, size_t rec, size_t hole, int n)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const unsigned char *s = src + (size_t)i * rec;
    unsigned char *d = dst + (size_t)i * (rec + hole) + 3;                 // odd alignment on purpose
    const size_t head = (16 - ((size_t)d & 15)) & 15;
    if ((size_t)lane < head) d[lane] = s[lane];
    const size_t body = (rec - head) / 16;
    const unsigned char *sb = s + head; uint4 *db = (uint4 *)(d + head);
    for (size_t k = lane; k < body; k += 64) { uint4 v; memcpy(&v, sb + 16 * k, 16); db[k] = v; }
    const size_t done = head + 16 * body;
    if (done + lane < rec) d[done + lane] = s[done + lane];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const size_t N = (size_t)1 << 30;
    unsigned char *h_in, *h_out, *d_a, *d_b;
    CHK(hipHostMalloc((void **)&h_in, N, hipHostMallocDefault)); CHK(hipHostMalloc((void **)&h_out, N + (64 << 20), hipHostMallocDefault));
    CHK(hipMalloc((void **)&d_a, N)); CHK(hipMalloc((void **)&d_b, N));
    memset(h_in, 1, N); memset(h_out, 2, N);
    CHK(hipMemset(d_a, 3, N)); CHK(hipMemset(d_b, 4, N));
    hipStream_t s1, s2; CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e1a, e1b, e2a, e2b; CHK(hipEventCreate(&e1a)); CHK(hipEventCreate(&e1b)); CHK(hipEventCreate(&e2a)); CHK(hipEventCreate(&e2b));
    unsigned char *hd_in = nullptr, *hd_out = nullptr;
    CHK(hipHostGetDevicePointer((void **)&hd_in, h_in, 0)); CHK(hipHostGetDevicePointer((void **)&hd_out, h_out, 0));
    const size_t n16 = N / 16;

    auto h2d_sdma = [&]() { CHK(hipMemcpyAsync(d_a, h_in, N, hipMemcpyHostToDevice, s1)); };
    auto d2h_sdma = [&]() { CHK(hipMemcpyAsync(h_out, d_b, N, hipMemcpyDeviceToHost, s2)); };
    int W = 64;
    auto h2d_shader = [&]() { hipLaunchKernelGGL(copy16, dim3(W), dim3(256), 0, s1, (const uint4 *)hd_in, (uint4 *)d_a, n16); };
    auto d2h_shader = [&]() { hipLaunchKernelGGL(copy16, dim3(W), dim3(256), 0, s2, (const uint4 *)d_b, (uint4 *)hd_out, n16); };
    const size_t rec = 26000, hole = 15300; const int nrec = (int)(N / (rec + hole));
    auto d2h_lines = [&]() { hipLaunchKernelGGL(copy_lines, dim3(nrec), dim3(64), 0, s2, (const unsigned char *)d_b, hd_out, rec, hole, nrec); };

    auto run = [&](const char *name, auto fa, auto fb, bool has_a, bool has_b, double bytes_a, double bytes_b) {
        for (int w = 0; w < 2; w++) { if (has_a) fa(); if (has_b) fb(); CHK(hipDeviceSynchronize()); }
        const int reps = 4;
        double ta = 0, tb = 0;
        const double t0 = now();
        for (int r = 0; r < reps; r++) {
            if (has_a) { CHK(hipEventRecord(e1a, s1)); fa(); CHK(hipEventRecord(e1b, s1)); }
            if (has_b) { CHK(hipEventRecord(e2a, s2)); fb(); CHK(hipEventRecord(e2b, s2)); }
            CHK(hipDeviceSynchronize());
            float ms;
            if (has_a) { CHK(hipEventElapsedTime(&ms, e1a, e1b)); ta += ms; }
            if (has_b) { CHK(hipEventElapsedTime(&ms, e2a, e2b)); tb += ms; }
        }
        const double wall = (now() - t0) / reps;
        printf("%-44s wall %7.2f ms", name, wall * 1e3);
        if (has_a) printf("   H2D %6.1f GB/s (%6.2f ms)", bytes_a / (ta / reps * 1e-3) / 1e9, ta / reps);
        if (has_b) printf("   D2H %6.1f GB/s (%6.2f ms)", bytes_b / (tb / reps * 1e-3) / 1e9, tb / reps);
        printf("\n"); fflush(stdout);
    };
    auto none = [&]() {};
    run("sdma H2D alone", h2d_sdma, none, true, false, N, 0);
    run("sdma D2H alone", none, d2h_sdma, false, true, 0, N);
    run("sdma H2D | sdma D2H", h2d_sdma, d2h_sdma, true, true, N, N);
    const int Ws[] = { 8, 16, 32, 64, 128, 256, 1024 };
    for (int w : Ws) {
        W = w; char nm[128];
        snprintf(nm, sizeof nm, "shader H2D alone, %d workgroups", w); run(nm, h2d_shader, none, true, false, N, 0);
        snprintf(nm, sizeof nm, "shader D2H alone, %d workgroups", w); run(nm, none, d2h_shader, false, true, 0, N);
    }
    for (int w : { 16, 64, 256 }) {
        W = w; char nm[128];
        snprintf(nm, sizeof nm, "sdma H2D | shader D2H (%d wg)", w); run(nm, h2d_sdma, d2h_shader, true, true, N, N);
        snprintf(nm, sizeof nm, "shader H2D (%d wg) | sdma D2H", w); run(nm, h2d_shader, d2h_sdma, true, true, N, N);
        snprintf(nm, sizeof nm, "shader H2D | shader D2H (%d wg each)", w); run(nm, h2d_shader, d2h_shader, true, true, N, N);
    }
    run("shader D2H, 26 kB lines into holes (1 wave/line)", none, d2h_lines, false, true, 0, (double)nrec * rec);
    run("sdma H2D | shader D2H lines", h2d_sdma, d2h_lines, true, true, N, (double)nrec * rec);
    return 0;
}
