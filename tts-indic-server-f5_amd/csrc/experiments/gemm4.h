// gemm4: stream-K version of the warp-specialised MFMA GEMM (gemm3.h) for gfx950.
//
// Why: at one utterance per GPU the DiT GEMMs have 176 / 352 / 528 tiles of 128 x 128 for 256 CUs.  A data-parallel
// launch runs whole tiles, so the slowest CU owns ceil(tiles / slots) of them (176 tiles leave 80 CUs idle, 528 tiles on
// 512 slots pay a whole extra round for 16 tiles), and the k-loop itself runs at the CU's L2 -> LDS rate (~30 B/clk),
// not at the MFMA rate.  Both bounds are per CU, so the cure is the same: give every CU the same number of k-steps.
//
//   * work = tiles x (K / 32) k-steps, cut into gridDim.x equal contiguous ranges (grid = number of CUs);
//   * a workgroup walks its range from the TOP down, one segment per tile it touches, k ascending inside a segment;
//   * a segment that does not reach the end of its tile's K is a partial sum: the four consumer waves store their
//     accumulators (MFMA register layout, 64 KiB per workgroup) to a workspace slot with sc1 (agent-scope write-through) stores
//     and publish a per-wave flag -- per-access coherence bits, not release / acquire fences: those flush / invalidate the
//     XCD's whole L2, which holds the A / W tiles every CU is streaming (measured: 2x slower GEMMs);
//   * the workgroup whose segment holds the tile's LAST k-step finishes the tile: it adds the slots of the lower-numbered
//     workgroups that hold the rest of that tile (sc1 loads) and runs the fused epilogue.
// Deadlock freedom: a partial is always the FIRST segment a workgroup computes and a finisher only waits for
// LOWER-numbered workgroups, which the dispatcher starts earlier and which never wait before publishing.
// Flags carry a launch epoch (host counter), so they are never reset.
//
// Pipeline as gemm3.h: waves 4-7 stream k-tiles by LDS-DMA into a ring, waves 0-3 read fragments and issue MFMAs; one raw
// s_barrier per k-step.  The ring runs straight across segment boundaries.  The epilogue slabs have their own 64 KiB of
// LDS (not aliased with the ring), so the producers keep prefetching the next segment while the consumers finish a tile.
#pragma once
#include "../gemm3.h"

struct StreamKWs {
    float* slots = nullptr;        // [grid][4 waves][64 regs][64 lanes] fp32
    unsigned* flags = nullptr;     // [grid][4] + [1] count of timed-out waits (must stay 0)
    unsigned epoch = 0;
    int grid = 0;
};

template <int NSPLIT>
struct Gemm4Cfg {
    static constexpr int NPL = NSPLIT == 2 ? 2 : 1;
    static constexpr int STAGE = NPL * (128 + 128) * 64;
    static constexpr int NST = NPL == 2 ? 3 : 6;   // 96 KiB ring either way
    static constexpr int RING = NST * STAGE;
    static constexpr int LDS = RING + 4 * 16384;   // + one 64 x 64 fp32 epilogue slab per consumer wave = 160 KiB, one workgroup per CU
};

template <int NSPLIT, int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm4_kernel(const GemmArgs p, float* __restrict__ sk_slots,
                                                                                                unsigned* sk_flags, unsigned sk_epoch, int tiles_n, int total_iters) {
    using C = Gemm4Cfg<NSPLIT>;
    constexpr int NPL = C::NPL, NST = C::NST, STAGE = C::STAGE;
    constexpr bool F16 = NSPLIT == 3;
    constexpr int TM = 2, TN = 2;
    constexpr int A_PLANE = 128 * 64, B_PLANE = 128 * 64;
    constexpr int P = STAGE / 1024 / 4;   // DMA pieces per producer wave per k-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K >> 5;
    const int c = blockIdx.x, G = gridDim.x;
    const int s = (int)((long long)c * total_iters / G), e = (int)((long long)(c + 1) * total_iters / G);
    const int total = e - s;
    if (total <= 0) return;

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        // (lambdas must not capture the by-value kernel argument struct: its address would force a scratch copy of all of it)
        const int pw = wave - 4;
        const __bf16 *a0 = p.A[0], *a1 = NPL == 2 ? p.A[1] : p.A[0], *w0 = p.W[0], *w1 = NPL == 2 ? p.W[1] : p.W[0];
        const int lda = p.lda, ldw = p.ldw;
        int cur_end = e, seg_k = 0, seg_k1 = 0;
        const char* gsrc[P];
        // one loop, one issue site: step f < 0 is the prologue (fill NST - 1 stages), step f >= 0 retires tile f and refills its
        // predecessor's stage.  No stateful lambdas: hipcc keeps their captured scalars and pointer arrays in scratch.
        for (int f = -(NST - 1); f < total; f++) {
            if (f >= 0) {
                const int newer = min(NST - 2, total - 1 - f);   // tiles issued after tile f that may stay in flight
                if (newer == NST - 2) wait_vmcnt<(NST - 2) * P>();
                else if (newer == 3) wait_vmcnt<3 * P>();
                else if (newer == 2) wait_vmcnt<2 * P>();
                else if (newer == 1) wait_vmcnt<P>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            }
            const int fi = f + NST - 1;   // flat index of the tile to issue now
            if (fi >= total) continue;
            if (seg_k == seg_k1) {   // open the next segment (descending tiles)
                const int tile = (cur_end - 1) / nk, tf = tile * nk, sb = max(s, tf);
                seg_k = sb - tf; seg_k1 = cur_end - tf; cur_end = sb;
                const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
#pragma unroll
                for (int j = 0; j < P; j++) {
                    const int off = (pw * P + j) * 1024;
                    const bool isA = off < NPL * A_PLANE;
                    const int rel = isA ? off : off - NPL * A_PLANE;
                    const int pl = rel / A_PLANE;
                    const int row = (rel - pl * A_PLANE) / 64 + (lane >> 2);
                    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
                    const __bf16* ap = pl ? a1 : a0;
                    const __bf16* wp = pl ? w1 : w0;
                    const __bf16* base = isA ? ap + (size_t)(m0 + row) * lda : wp + (size_t)(n0 + row) * ldw;
                    gsrc[j] = reinterpret_cast<const char*>(base + chunk * 8);
                }
            }
            char* dst = smem + (fi % NST) * STAGE + pw * (P * 1024);
#pragma unroll
            for (int j = 0; j < P; j++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[j] + (size_t)seg_k * 64),
                                                 (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
            seg_k++;
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[TM][TN];
#define G4_ZERO_ACC()                                                     \
    _Pragma("unroll") for (int i = 0; i < TM; i++)                        \
        _Pragma("unroll") for (int j = 0; j < TN; j++)                    \
            _Pragma("unroll") for (int g = 0; g < 16; g++) acc[i][j][g] = 0.0f;
    G4_ZERO_ACC();

    bf16x8 fa[2][NPL][TM], fb[2][NPL][TN];
    auto read_frags = [&](int buf, const char* st, int sh) {
        const int chunk = sh * 2 + fh;
#pragma unroll
        for (int pl = 0; pl < NPL; pl++) {
#pragma unroll
            for (int i = 0; i < TM; i++)
                fa[buf][pl][i] = *reinterpret_cast<const bf16x8*>(st + pl * A_PLANE + lds_off2(wm * 64 + i * 32 + fr, chunk));
#pragma unroll
            for (int j = 0; j < TN; j++)
                fb[buf][pl][j] = *reinterpret_cast<const bf16x8*>(st + NPL * A_PLANE + pl * B_PLANE + lds_off2(wn * 64 + j * 32 + fr, chunk));
        }
    };
    auto mfma_frags = [&](int buf) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) {
                if (NSPLIT == 2) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][1][i], fb[buf][0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][0][i], fb[buf][1][j], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = mfma_32x32x16<F16>(fa[buf][0][i], fb[buf][0][j], acc[i][j]);
            }
    };

    // segment bookkeeping (same walk as the producers)
    int cur_end = e, tile = 0, k0 = 0, k1 = 0, rem = 0;
#define G4_OPEN_SEGMENT()                                              \
    {                                                                  \
        tile = (cur_end - 1) / nk;                                     \
        const int tf_ = tile * nk, sb_ = max(s, tf_);                  \
        k0 = sb_ - tf_; k1 = cur_end - tf_; cur_end = sb_; rem = k1 - k0; \
    }
    float* slab = reinterpret_cast<float*>(smem + C::RING) + wave * 4096;
    f32x4* my_slot = reinterpret_cast<f32x4*>(sk_slots) + ((size_t)c * 4 + wave) * 1024;   // 64 regs x 64 lanes fp32 = 1024 x 16 B
    G4_OPEN_SEGMENT();
    __builtin_amdgcn_s_barrier();                         // B_0: tile 0 landed
    read_frags(0, smem, 0);
    for (int f = 0; f < total; f++) {
        const char* st = smem + (f % NST) * STAGE;
        read_frags(1, st, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_frags(0);
        __builtin_amdgcn_sched_barrier(0);
        if (f + 1 < total) {
            // every LDS read of tile f has returned (issued 12 / 4 MFMAs ago): the producers may refill its stage
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // B_{f+1}: tile f+1 landed
            read_frags(0, smem + ((f + 1) % NST) * STAGE, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        mfma_frags(1);
        __builtin_amdgcn_sched_barrier(0);
        if (--rem == 0) {
            {   // ---- finish the segment (inline, not a lambda: see the note on the producers)
                const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
                if (k1 < nk) {
                    // partial sum of tile `tile`: k-steps [k0, k1); the finisher is a higher-numbered workgroup
#pragma unroll
                    for (int i = 0; i < TM; i++)
#pragma unroll
                        for (int j = 0; j < TN; j++)
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                                // sc1: agent-scope write-through store -- visible to the other XCDs without flushing this XCD's whole L2
                                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(my_slot + ((i * TN + j) * 4 + q) * 64 + lane), "v"(v) : "memory");
                            }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the slot is written through before the flag goes up
                    __hip_atomic_store(sk_flags + c * 4 + wave, sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                if (k0 > 0) {
                    // the rest of this tile's K lives in workgroups c-1, c-2, ... down to the one that holds the tile's first k-step
                    const int tf = tile * nk;
                    for (int cc = c - 1; cc >= 0; cc--) {
                        const int scc = (int)((long long)cc * total_iters / G);
                        // bounded spin (~0.1 s): a protocol bug must end as a counted error (sk_flags[4 G]), never as a hung GPU
                        int spins = 0;
                        while (__hip_atomic_load(sk_flags + cc * 4 + wave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk_epoch && ++spins < (1 << 20))
                            __builtin_amdgcn_s_sleep(2);
                        if (spins >= (1 << 20) && lane == 0) atomicAdd(sk_flags + 4 * G, 1u);
                        const f32x4* slot = reinterpret_cast<const f32x4*>(sk_slots) + ((size_t)cc * 4 + wave) * 1024;
                        // sc1 loads: agent-scope coherent reads of the peer's slot (no invalidate of this XCD's L2, which holds the A / W tiles)
                        // NB regs x 4 dwords of loads in flight per batch: a 32 x 64 half tile (one plane), 8 rows of a 32 x 32 tile under
                        // split-bf16 register pressure (its fragments take 64 registers)
                        constexpr int NB = NPL == 2 ? 2 : 8;
#pragma unroll
                        for (int b0 = 0; b0 < TM * TN * 4; b0 += NB) {
                            f32x4 pv[NB];
#pragma unroll
                            for (int t = 0; t < NB; t++)
                                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(pv[t]) : "v"(slot + (b0 + t) * 64 + lane) : "memory");
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                            for (int t = 0; t < NB; t++) {
                                const int ij = (b0 + t) / 4, q = (b0 + t) % 4;
                                acc[ij / TN][ij % TN][4 * q] += pv[t][0]; acc[ij / TN][ij % TN][4 * q + 1] += pv[t][1];
                                acc[ij / TN][ij % TN][4 * q + 2] += pv[t][2]; acc[ij / TN][ij % TN][4 * q + 3] += pv[t][3];
                            }
                        }
                        if (scc <= tf) break;
                    }
                }
                // fused epilogue through the wave's private slab (no workgroup barrier: the slab is not part of the ring)
                gemm_epilogue<EPI, TM, TN, false>(p, acc, slab, m0 + wm * 64, n0 + wn * 64, n0, lane);
                }
            }
            if (f + 1 < total) {
                G4_ZERO_ACC();
                G4_OPEN_SEGMENT();
            }
        }
    }
#undef G4_ZERO_ACC
#undef G4_OPEN_SEGMENT
}

template <int NSPLIT, int EPI>
static hipError_t launch_gemm4_t(const GemmArgs& a, int m_pad, int n_pad, StreamKWs& ws, hipStream_t st) {
    using C = Gemm4Cfg<NSPLIT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4_kernel<NSPLIT, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles_n = n_pad / 128, tiles = (m_pad / 128) * tiles_n, total = tiles * (a.K >> 5);
    const int grid = total < ws.grid ? total : ws.grid;
    ws.epoch++;
    hipLaunchKernelGGL((gemm4_kernel<NSPLIT, EPI>), dim3(grid), dim3(512), C::LDS, st, a, ws.slots, ws.flags, ws.epoch, tiles_n, total);
    return hipGetLastError();
}

static int streamk_ws_init(StreamKWs& ws) {
    if (ws.slots) return 0;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    ws.grid = prop.multiProcessorCount;
    if (hipMalloc((void**)&ws.slots, (size_t)ws.grid * 4 * 64 * 64 * sizeof(float)) != hipSuccess) return -1;
    if (hipMalloc((void**)&ws.flags, ((size_t)ws.grid * 4 + 1) * sizeof(unsigned)) != hipSuccess) return -1;   // + the time-out counter
    if (hipMemset(ws.flags, 0, ((size_t)ws.grid * 4 + 1) * sizeof(unsigned)) != hipSuccess) return -1;
    ws.epoch = 0;
    return 0;
}
static void streamk_ws_free(StreamKWs& ws) {
    if (ws.slots) (void)hipFree(ws.slots);
    if (ws.flags) (void)hipFree(ws.flags);
    ws = StreamKWs();
}
