// tools/fillrate.hip -- L2 -> LDS operand fill-rate microbenchmark for GEMM-shaped streams on MI355X (diagnostics, not product).
//
// Every workgroup streams the A row panel and the W row panel of "its" output tile through an LDS ring exactly like the
// GEMM's loader does (1 KiB LDS-DMA pieces, counted vmcnt, one barrier per k-step), but nothing is computed: the time is
// the operand fill alone.  What it answers:
//   * ROWB: bytes of one tile row per k-step: 64 (BK = 32 fp16, half a cache line), 128 (BK = 64), 256 (BK = 128);
//   * tile shape / grid: 128 x 128 (176 tiles at the C2 out-projection) vs the exact-fit 176 x {64,128,192} (256 tiles);
//   * blockIdx -> tile mapping: row-major vs XCD-blocked (blocks b and b + 8 share an XCD);
//   * loader waves per workgroup and bytes in flight.
// A is L2 / MALL resident (re-used every launch), W cycles through a pool larger than the Infinity Cache (HBM-cold, like the
// weights of consecutive layers in the real forward).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/fillrate tools/fillrate.hip ; run: tools/fillrate > gpurun_out/fillrate.txt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct FillArgs {
    const char* A; const char* W;
    int lda, ldw;          // row strides in bytes
    int kbytes;            // bytes of K per row
    int tiles_m, tiles_n;  // grid of tiles
    int map;               // 0 row-major (n fastest), 1 XCD-blocked: xcd = b % 8 owns a contiguous block of tiles
    unsigned long long* stamps;   // [grid][2] s_memrealtime at start / end
};

// MODE 0: LDS-DMA (global_load_lds_dwordx4); MODE 1: global_load_dwordx4 -> registers -> ds_write_b128 one step later
template <int BM, int BN, int ROWB, int NW, int DEPTH /* k-steps in flight */, int MODE>
__global__ __launch_bounds__(NW * 64) void fill_kernel(const FillArgs p) {
    constexpr int PIECES = (BM + BN) * ROWB / 1024;
    constexpr int P = (PIECES + NW - 1) / NW;
    constexpr int RPP = 1024 / ROWB;        // rows per piece
    constexpr int CPR = ROWB / 16;          // 16-byte chunks per row
    constexpr int STAGE = PIECES * 1024;
    constexpr int NST = DEPTH + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x, tm, tn;
    const int nt = p.tiles_m * p.tiles_n;
    if (p.map == 1) {
        // blocks b, b + 8, ... share an XCD: give that XCD the contiguous tile range [xcd * nt / 8, (xcd + 1) * nt / 8)
        const int xcd = b & 7, idx = b >> 3, per = nt >> 3;
        b = xcd * per + idx;
    }
    if (p.map == 2) {   // XCD-blocked, m fastest inside the XCD's range (W panel shared by consecutive blocks)
        const int xcd = b & 7, idx = b >> 3, per = nt >> 3;
        b = xcd * per + idx;
        tm = b % p.tiles_m; tn = b / p.tiles_m;
    } else {
        tm = b / p.tiles_n; tn = b % p.tiles_n;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const char* src[P];
#pragma unroll
    for (int j = 0; j < P; j++) {
        const int pc = wave + j * NW;          // piece index (interleaved over the waves)
        const int row = pc * RPP + lane / CPR;
        const int chunk = lane % CPR;
        const bool isA = row < BM;
        const char* base = isA ? p.A + (size_t)(m0 + row) * p.lda : p.W + (size_t)(n0 + row - BM) * p.ldw;
        src[j] = base + chunk * 16;
    }
    const int nk = p.kbytes / ROWB;
    unsigned long long t0 = 0, t1 = 0;
    if (tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 regs[MODE == 1 ? P : 1];
    auto issue = [&](int kt) {
        char* dst = smem + (kt % NST) * STAGE;
#pragma unroll
        for (int j = 0; j < P; j++) {
            const int pc = wave + j * NW;
            if (pc < PIECES) {
                if constexpr (MODE == 0)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (size_t)kt * ROWB),
                                                     (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
                else
                    regs[j] = *reinterpret_cast<const u32x4*>(src[j] + (size_t)kt * ROWB);
            }
        }
    };
    if constexpr (MODE == 0) {
#pragma unroll
        for (int t = 0; t < DEPTH; t++)
            if (t < nk) issue(t);
        for (int kt = 0; kt < nk; kt++) {
            if (kt + DEPTH <= nk) wait_vmcnt<(DEPTH - 1) * P>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + DEPTH < nk) issue(kt + DEPTH);
        }
    } else {
        // register staging: one step in flight in registers, written to LDS after the next step's loads are issued
        issue(0);
        for (int kt = 0; kt < nk; kt++) {
            u32x4 cur[P];
#pragma unroll
            for (int j = 0; j < P; j++) cur[j] = regs[j];
            if (kt + 1 < nk) issue(kt + 1);
            char* dst = smem + (kt % NST) * STAGE;
#pragma unroll
            for (int j = 0; j < P; j++) {
                const int pc = wave + j * NW;
                if (pc < PIECES) *reinterpret_cast<u32x4*>(dst + pc * 1024 + lane * 16) = cur[j];
            }
            __builtin_amdgcn_s_barrier();
        }
    }
    __syncthreads();
    if (tid == 0) {
        t1 = __builtin_amdgcn_s_memrealtime();
        // keep the LDS contents alive
        unsigned v = *reinterpret_cast<volatile unsigned*>(smem + (lane * 4));
        p.stamps[2 * blockIdx.x] = t0;
        p.stamps[2 * blockIdx.x + 1] = t1 + (v & 0);
    }
}

struct Result { double span_us, med_us, evt_us; };

template <int BM, int BN, int ROWB, int NW, int DEPTH, int MODE>
static Result run(const char* A, const std::vector<char*>& Wpool, int M, int N, int K2 /* bytes */, int map, int iters) {
    constexpr int PIECES = (BM + BN) * ROWB / 1024;
    constexpr int LDS = (DEPTH + 1) * PIECES * 1024;
    static_assert(LDS <= 160 * 1024, "ring too large");
    FillArgs a;
    a.A = A; a.lda = K2; a.ldw = K2; a.kbytes = K2; a.tiles_m = M / BM; a.tiles_n = N / BN; a.map = map;
    const int grid = a.tiles_m * a.tiles_n;
    unsigned long long* st;
    CHECK(hipMalloc(&st, sizeof(unsigned long long) * 2 * grid));
    a.stamps = st;
    auto kern = fill_kernel<BM, BN, ROWB, NW, DEPTH, MODE>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    std::vector<double> spans, meds;
    std::vector<unsigned long long> h(2 * grid);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double evt = 0;
    for (int it = 0; it < iters + 3; it++) {
        a.W = Wpool[it % Wpool.size()];
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), LDS, 0, a);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        if (it < 3) continue;
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); evt += ms * 1e3;
        CHECK(hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
        unsigned long long lo = ~0ull, hi = 0;
        std::vector<double> d(grid);
        for (int g = 0; g < grid; g++) { lo = std::min(lo, h[2 * g]); hi = std::max(hi, h[2 * g + 1]); d[g] = (h[2 * g + 1] - h[2 * g]) * 0.01; }
        std::sort(d.begin(), d.end());
        spans.push_back((hi - lo) * 0.01); meds.push_back(d[grid / 2]);
    }
    std::sort(spans.begin(), spans.end()); std::sort(meds.begin(), meds.end());
    CHECK(hipFree(st));
    return {spans[spans.size() / 2], meds[meds.size() / 2], evt / iters};
}

template <int BM, int BN, int ROWB, int NW, int DEPTH, int MODE>
static void report(const char* tag, const char* A, const std::vector<char*>& Wpool, int M, int N, int K, int map) {
    Result r = run<BM, BN, ROWB, NW, DEPTH, MODE>(A, Wpool, M, N, K * 2, map, 40);
    const double bytes_wg = (double)(BM + BN) * K * 2;
    const int grid = (M / BM) * (N / BN);
    const double gbs = bytes_wg / (r.med_us * 1e-6) / 1e9;
    printf("%-34s M%5d N%5d K%5d tile %3dx%3d rowB %3d waves %d depth %d mode %d map %d | grid %3d | WG bytes %7.0f KB | WG median %6.2f us = %6.1f GB/s/CU = %5.1f B/clk@2.4 | span %6.2f us | event %6.2f us\n",
           tag, M, N, K, BM, BN, ROWB, NW, DEPTH, MODE, map, grid, bytes_wg / 1024, r.med_us, gbs, gbs / 2.4, r.span_us, r.evt_us);
    fflush(stdout);
}

int main() {
    const int M = 2816, KMAX = 2048, NMAX = 3072;
    char* A;
    CHECK(hipMalloc(&A, (size_t)M * KMAX * 2));
    CHECK(hipMemset(A, 1, (size_t)M * KMAX * 2));
    // W pool: 48 x 12 MiB = 576 MiB > Infinity Cache
    std::vector<char*> pool(48);
    for (auto& w : pool) { CHECK(hipMalloc(&w, (size_t)NMAX * KMAX * 2)); CHECK(hipMemset(w, 2, (size_t)NMAX * KMAX * 2)); }
    std::vector<char*> hot(1, pool[0]);   // same W every launch: L2 / MALL hot

    printf("# out projection  (M 2816, N 1024, K 1024)\n");
    report<128, 128, 64, 4, 3, 0>("128x128 BK32 (round-1 gemm3)", A, pool, M, 1024, 1024, 0);
    report<128, 128, 64, 4, 3, 0>("128x128 BK32 hot W", A, hot, M, 1024, 1024, 0);
    report<128, 128, 128, 4, 2, 0>("128x128 BK64 depth2", A, pool, M, 1024, 1024, 0);
    report<128, 128, 128, 4, 3, 0>("128x128 BK64 depth3", A, pool, M, 1024, 1024, 0);
    report<128, 128, 128, 8, 3, 0>("128x128 BK64 depth3 8 waves", A, pool, M, 1024, 1024, 0);
    report<128, 128, 256, 4, 1, 0>("128x128 BK128 depth1", A, pool, M, 1024, 1024, 0);
    report<128, 128, 128, 4, 3, 0>("128x128 BK64 depth3 hot W", A, hot, M, 1024, 1024, 0);
    report<128, 128, 128, 4, 1, 1>("128x128 BK64 reg-staged", A, pool, M, 1024, 1024, 0);
    report<128, 128, 128, 8, 1, 1>("128x128 BK64 reg-staged 8 waves", A, pool, M, 1024, 1024, 0);
    report<176, 64, 128, 4, 3, 0>("176x64 BK64 depth3 map0", A, pool, M, 1024, 1024, 0);
    report<176, 64, 128, 4, 3, 0>("176x64 BK64 depth3 map1", A, pool, M, 1024, 1024, 1);
    report<176, 64, 128, 4, 3, 0>("176x64 BK64 depth3 map2", A, pool, M, 1024, 1024, 2);
    report<176, 64, 128, 8, 3, 0>("176x64 BK64 depth3 map1 8w", A, pool, M, 1024, 1024, 1);
    report<176, 64, 128, 4, 4, 0>("176x64 BK64 depth4 map1", A, pool, M, 1024, 1024, 1);
    report<176, 64, 256, 4, 1, 0>("176x64 BK128 depth1 map1", A, pool, M, 1024, 1024, 1);
    report<176, 64, 64, 4, 4, 0>("176x64 BK32 depth4 map1", A, pool, M, 1024, 1024, 1);
    printf("# FF1  (N 2048, K 1024)\n");
    report<128, 128, 64, 4, 3, 0>("128x128 BK32 (round-1 gemm3)", A, pool, M, 2048, 1024, 0);
    report<176, 128, 128, 4, 3, 0>("176x128 BK64 depth3 map0", A, pool, M, 2048, 1024, 0);
    report<176, 128, 128, 4, 3, 0>("176x128 BK64 depth3 map1", A, pool, M, 2048, 1024, 1);
    report<176, 128, 128, 4, 3, 0>("176x128 BK64 depth3 map2", A, pool, M, 2048, 1024, 2);
    report<176, 128, 128, 8, 3, 0>("176x128 BK64 depth3 map1 8w", A, pool, M, 2048, 1024, 1);
    printf("# QKV  (N 3072, K 1024)\n");
    report<128, 128, 64, 4, 3, 0>("128x128 BK32 (round-1 gemm3)", A, pool, M, 3072, 1024, 0);
    report<176, 192, 128, 4, 2, 0>("176x192 BK64 depth2 map0", A, pool, M, 3072, 1024, 0);
    report<176, 192, 128, 4, 2, 0>("176x192 BK64 depth2 map1", A, pool, M, 3072, 1024, 1);
    report<176, 192, 128, 4, 2, 0>("176x192 BK64 depth2 map2", A, pool, M, 3072, 1024, 2);
    report<176, 192, 128, 8, 2, 0>("176x192 BK64 depth2 map1 8w", A, pool, M, 3072, 1024, 1);
    printf("# FF2  (N 1024, K 2048)\n");
    report<128, 128, 64, 4, 3, 0>("128x128 BK32 (round-1 gemm3)", A, pool, M, 1024, 2048, 0);
    report<176, 64, 128, 4, 3, 0>("176x64 BK64 depth3 map1", A, pool, M, 1024, 2048, 1);
    report<176, 64, 128, 4, 3, 0>("176x64 BK64 depth3 map2", A, pool, M, 1024, 2048, 2);
    report<176, 64, 256, 4, 1, 0>("176x64 BK128 depth1 map1", A, pool, M, 1024, 2048, 1);
    return 0;
}
