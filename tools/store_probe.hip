// tools/store_probe.hip -- what a kernel boundary costs when the kernel has just written N bytes, by store flavour (diagnostics, MI355X).
//
// Question (round 3): an attention launch at C2 costs 2.7 us when its workgroups return at once, 10.3 us with prologue + epilogue and no KV loop;
// the per-wave stamps account for 4.4 us of the difference.  Is the rest the end-of-kernel write-back of the dirty output lines (each XCD's L2 is
// write-back and is flushed when a kernel ends), and would stores that write through (sc1 / sc0 sc1 / nt) move that cost into the kernel body?
//
// producer: 256 workgroups x 256 threads write `bytes` with 16-byte stores of one flavour; consumer: reads them back (sum), so that a
// producer -> consumer chain pays for data that has to reach the consumer's XCD.  Times (HIP events, back-to-back launches on one stream):
//   P only, P -> C pairs, for bytes in {0, 1.4 MB, 5.8 MB, 11.5 MB, 23 MB} and flavours {plain, nt, sc1, sc0 sc1}.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/store_probe tools/store_probe.hip ; run: tools/store_probe > gpurun_out/store_probe.txt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int FLAVOUR>
__device__ __forceinline__ void store16(char* p, u32x4 v) {
    if (FLAVOUR == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else if (FLAVOUR == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else if (FLAVOUR == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// every workgroup writes a contiguous slice; a workgroup's threads cover 4 KiB per step
template <int FLAVOUR>
__global__ __launch_bounds__(256) void producer(char* out, size_t bytes, unsigned seed) {
    const size_t per = bytes / gridDim.x;
    char* base = out + (size_t)blockIdx.x * per;
    const u32x4 v = {seed, seed + 1, seed + 2, threadIdx.x};
    for (size_t off = (size_t)threadIdx.x * 16; off < per; off += 256 * 16) store16<FLAVOUR>(base + off, v);
}

// the consumer of workgroup b reads the slice that workgroup (b + shift) % grid wrote: shift = 0 same XCD, shift = 1 the next XCD
__global__ __launch_bounds__(256) void consumer(const char* in, size_t bytes, int shift, unsigned* sink) {
    const size_t per = bytes / gridDim.x;
    const char* base = in + (size_t)((blockIdx.x + shift) % gridDim.x) * per;
    unsigned acc = 0;
    for (size_t off = (size_t)threadIdx.x * 16; off < per; off += 256 * 16) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(base + off);
        acc += v[0] ^ v[3];
    }
    if (acc == 0x12345678u) sink[0] = acc;   // (never true: keeps the loads)
}

template <int FLAVOUR>
static void run(char* buf, unsigned* sink, size_t bytes, const char* name) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 300;
    float ms_p = 0, ms_pc0 = 0, ms_pc1 = 0;
    for (int mode = 0; mode < 3; mode++) {
        for (int it = -20; it < iters; it++) {
            if (it == 0) CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(producer<FLAVOUR>, dim3(256), dim3(256), 0, 0, buf, bytes, (unsigned)it);
            if (mode) hipLaunchKernelGGL(consumer, dim3(256), dim3(256), 0, 0, buf, bytes, mode == 1 ? 0 : 1, sink);
        }
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        (mode == 0 ? ms_p : mode == 1 ? ms_pc0 : ms_pc1) = ms;
    }
    printf("%-8s %8.2f MB | producer alone %7.2f us | producer + consumer (same XCD slice) %7.2f us | (next XCD's slice) %7.2f us\n", name, bytes / 1e6,
           ms_p * 1e3 / iters, ms_pc0 * 1e3 / iters, ms_pc1 * 1e3 / iters);
    fflush(stdout);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main() {
    char* buf; unsigned* sink;
    CHECK(hipMalloc(&buf, 64u << 20)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 0, 64u << 20));
    const size_t sizes[] = {0, 1441792, 5767168, 11534336, 23068672};   // 0 | 2816 x 256 fp16 | 2816 x 1024 fp16 | 2816 x 1024 fp32 | 2816 x 2048 fp32
    for (size_t b : sizes) {
        run<0>(buf, sink, b, "plain");
        run<1>(buf, sink, b, "nt");
        run<2>(buf, sink, b, "sc1");
        run<3>(buf, sink, b, "sc0 sc1");
        printf("\n");
    }
    return 0;
}
