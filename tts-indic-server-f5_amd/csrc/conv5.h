// Sliding-window implicit-GEMM Conv1d over channel-last rows for gfx950 (the BigVGAN convolutions: k = 3 / 7 / 11, dilation 1 / 3 / 5,
// and the 3-tap form of the transposed convolutions):
//
//     out[t][n] = bias[n] + sum_tap sum_c A[t + (tap - center) * dil][c] * W[n][tap][c]  (+ res[t][n]),   A rows outside the sequence = 0
//
// gemm.h runs this as a GEMM whose A tile is re-read from L2 for every tap.  Here the rows a 256-row output tile depends on -- the
// WINDOW, 256 + 2 * center * dil rows of one channel chunk -- are brought into LDS ONCE per chunk, and every tap reads its MFMA fragments
// from that image at a row offset; only the weight tile of the (chunk, tap) streams through a ring.  Per k-step the L2 -> LDS fill drops
// from (256 + BN) rows to BN rows, and at the 24-96-channel stages the activation planes are read once instead of once per tap.
//
//  * 8 waves: waves 0-3 consume (64 output rows x all BN columns each, v_mfma_f32_32x32x16, fp32 accumulators in registers),
//    waves 4-7 load.  Per k-step (one tap of one chunk) the loaders DMA the weight tile (global_load_lds_dwordx4, source-side swizzle)
//    into a 5-6 stage ring; once per chunk they register-stage the next chunk's window (predicated 16-byte loads: rows outside
//    [sequence start, sequence start + valid) are zero, which is the convolution's zero padding) into the other window buffer.
//  * One raw s_barrier per k-step joins both groups: behind B_kt the weight tile of k-step kt has landed (counted vmcnt in the loaders),
//    the window of its chunk is written (lgkmcnt(0) in the loaders), and every read of k-step kt - 1 has returned (the consumers place
//    B_kt in the middle of k-step kt - 1, behind lgkmcnt(0)), so its stage and -- at a chunk boundary -- the window buffer of the chunk
//    before are free.  Fragments are double-buffered by half k-steps, so reads are always one half ahead of the MFMAs that use them.
//  * LDS rows are ROWB = 128 bytes (64 channels: one fp16 plane) or 64 bytes (32 channels: split bf16, or fp16 at channel counts that
//    are not a multiple of 64); 16-byte chunks are XOR-swizzled with (row >> 1) & 7 resp. (row >> 2) & 3, which keeps the four 16-lane
//    groups of a ds_read_b128 conflict-free for ANY row offset (a tap shift only adds a constant to the key sequence).
//  * Epilogue: bias, fp32 residual, fp32 store straight from the accumulators (a lane holds one column of 16 rows; the 32 lanes of a
//    half-wave write 128 contiguous bytes per row).
//  * NPL = 2: split bf16 operands, three MFMAs per fragment pair (hi*hi + hi*lo + lo*hi); NPL = 1: one plane, fp16 (F16) or bf16.
#pragma once
#include "common.h"
#include "gemm_epilogue.h"

template <int ROWB>
F5_DEVICE int c5_off(int row, int c16) {
    if constexpr (ROWB == 128) return row * 128 + ((c16 ^ ((row >> 1) & 7)) << 4);
    else return row * 64 + ((c16 ^ ((row >> 2) & 3)) << 4);
}

// RBW = 32-row blocks per consumer wave: 2 -> 256-row tiles (the BigVGAN stages, whose sequences are multiples of 256 rows), 1 -> 128-row
// tiles (the DiT's conv_pos_embed: sequences are padded to 128 rows, so a 128-row tile never straddles two of them)
template <int NPL, int ROWB, int NB, int RBW = 2>
struct Conv5Cfg {
    static constexpr int BM = 128 * RBW, BN = NB * 32, HMAX = 25;
    static constexpr int CKC = ROWB / 2;                  // channels per chunk
    static constexpr int CPR = ROWB / 16;                 // 16-byte chunks per LDS row
    static constexpr int WST = NPL * BN * ROWB;           // one weight stage (all planes)
    static constexpr int PW = WST / 1024;                 // 1 KiB DMA pieces per stage
    static constexpr int P_HI = (PW + 3) / 4, P_LO = PW / 4;
    static constexpr int WIN_MAX = ((BM + 2 * HMAX + 7) & ~7) * CPR * NPL;   // 16-byte pieces of the largest window
    static constexpr int WV = (WIN_MAX + 255) / 256;      // window pieces per loader lane
    // Weight ring depth: a tile issued behind barrier B_kt is needed at B_{kt + NST - 1}.  With 3 stages the k-loop ran at ~1.1 us per
    // k-step whatever the tap count (0.43 us of MFMA work): the two k-steps of lead do not cover the L2 / Infinity-Cache latency of a
    // weight slice that shares the 4 MiB L2 with the streaming windows.  5 stages of 16 KiB still fit beside two 39 KiB windows.
    static constexpr int NST = WST >= 16384 ? 5 : 3;   // (the narrow tiles have 3-33 k-steps and want LDS for a second workgroup instead)
    static constexpr int WAVES_PER_EU = NB <= 2 ? 4 : 2;   // narrow tiles: 64 accumulator registers, two workgroups per CU
    static int win_rows(int halo) { return (BM + 2 * halo + 7) & ~7; }
    // Narrow 256-row tiles keep ONE window buffer even with several channel chunks (the loaders rewrite it between two extra barriers at
    // a chunk boundary): two workgroups then fit a CU, and the other one covers the gap.  Every other tile double-buffers the window.
    static constexpr bool SINGLE_WIN = NB <= 2 && RBW == 2;
    static int win_bufs(int nchunks) { return nchunks > 1 && !SINGLE_WIN ? 2 : 1; }
    static int lds_bytes(int halo, int nchunks) { return win_bufs(nchunks) * win_rows(halo) * ROWB * NPL + NST * WST; }
};

// ABL = 1 (diagnostics, only reached with GemmArgs::stamps set): s_memtime stamps of wave 0 (consumer) and wave 4 (loader) of every
// workgroup into p.stamps[block][16]: consumer 0 start, 1 loop begin, 2 loop end, 3 cycles spent at barriers, 4 end;
// loader 8 start, 9 loop begin, 10 loop end, 11 cycles in the counted waits, 12 at barriers, 13 in window stores, 14 in issue + fetch.
template <int NPL, int ROWB, int NB, bool F16, int ABL = 0, int RBW = 2>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(Conv5Cfg<NPL, ROWB, NB, RBW>::WAVES_PER_EU, Conv5Cfg<NPL, ROWB, NB, RBW>::WAVES_PER_EU))) void conv5_kernel(const GemmArgs p, const int tiles_m, const int taps, const int halo, const int wrows) {
    using C = Conv5Cfg<NPL, ROWB, NB, RBW>;
    constexpr int BM = C::BM, BN = C::BN, NST = C::NST, CKC = C::CKC, CPR = C::CPR, WST = C::WST, PW = C::PW, P_HI = C::P_HI, P_LO = C::P_LO, WV = C::WV;
    constexpr int KS = ROWB / 32;                              // 16-deep MFMA k-substeps per k-step
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-blocked tile order, column tile slow: the 32 CUs of an XCD work on one or two column tiles, whose weights stay in that L2
    const int nblk = gridDim.x;
    const int t = (nblk & 7) ? (int)blockIdx.x : ((int)blockIdx.x & 7) * (nblk >> 3) + ((int)blockIdx.x >> 3);
    const int tn = t / tiles_m, tm = t - tn * tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int c_pad = p.K / taps;                              // channels per tap in the weight's K layout (== lda for a dense convolution)
    const int a_col0 = tn * p.conv_group_cols;                 // grouped convolution: column tile = group, its input channels start here
    const int nchunks = c_pad / CKC, nk = nchunks * taps;
    const int dil = p.conv_dil > 1 ? p.conv_dil : 1;
    const int win_plane = wrows * ROWB;                        // bytes of one plane of one window buffer
    const int win_buf = win_plane * NPL;
    char* const win0 = smem;
    constexpr bool SINGLE_WIN = C::SINGLE_WIN;
    const int wsel = SINGLE_WIN ? 0 : 1;                       // window of chunk c lives in buffer c & wsel
    char* const wst0 = smem + (nchunks > 1 && !SINGLE_WIN ? 2 : 1) * win_buf;
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define C5_NOW() ((unsigned long long)__builtin_amdgcn_s_memtime())
    if constexpr (ABL) ts[0] = C5_NOW();

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------------ loader waves
        const int pw = wave - 4, ll = tid - 256;
        const int mine = (PW - pw + 3) >> 2;                   // weight pieces pw, pw + 4, ... of every k-step
        // the tile lies in one sequence (BM divides the sequence pitch / padding): its bounds are those of its first row
        const int sstart = p.row_seq_start ? p.row_seq_start[m0] : (m0 / p.seq_pitch) * p.seq_pitch;
        const int send = p.row_seq_start ? p.row_seq_end[m0] : sstart + p.seq_valid;
        const char* wsrc[P_HI];
#pragma unroll
        for (int j = 0; j < P_HI; j++) {
            const int pc = pw + 4 * j;
            const int row = pc * (1024 / ROWB) + lane / CPR;   // row of the stage image: plane-major, BN rows per plane
            const int pl = row / BN, nl = row - pl * BN;
            const int phys = lane % CPR;
            const int logical = ROWB == 128 ? (phys ^ ((nl >> 1) & 7)) : (phys ^ ((nl >> 2) & 3));
            wsrc[j] = reinterpret_cast<const char*>(p.W[pl < NPL ? pl : 0] + (size_t)(n0 + nl) * p.ldw + logical * 8);
        }
        auto issue_w = [&](int kt) {
            const int chunk = kt / taps, tap = kt - chunk * taps;
            const size_t koff = ((size_t)tap * c_pad + (size_t)chunk * CKC) * 2;
            char* dst = wst0 + (kt % NST) * WST + pw * 1024;
#pragma unroll
            for (int j = 0; j < P_HI; j++)
                if (j < P_LO || pw + 4 * j < PW)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[j] + koff),
                                                     (__attribute__((address_space(3))) void*)(dst + j * 4096), 16, 0, 0);
        };
        // Counted wait.  vmcnt retires in issue order, so "the weight tile of k-step kt has landed" == "at most the operations issued
        // after it are outstanding": `newer` weight groups of `mine` DMAs each, plus the WV window loads while they are younger than the
        // group waited for (the first NST - 1 barriers after a fetch; once a window has been stored the compiler's full drain in front of
        // its LDS writes has completed everything older, so the count only ever matters for groups issued after the last store).
        // The counts must be exact, so the window loads are unconditional (clamped address, value zeroed by select afterwards): a
        // predicated load that a fully masked wave skips would make the wait too lenient, i.e. a race.
        auto wait_outstanding = [&](int n) {
            switch (n) {
                case 0: wait_vmcnt<0>(); break;   case 1: wait_vmcnt<1>(); break;   case 2: wait_vmcnt<2>(); break;   case 3: wait_vmcnt<3>(); break;
                case 4: wait_vmcnt<4>(); break;   case 5: wait_vmcnt<5>(); break;   case 6: wait_vmcnt<6>(); break;   case 7: wait_vmcnt<7>(); break;
                case 8: wait_vmcnt<8>(); break;   case 9: wait_vmcnt<9>(); break;   case 10: wait_vmcnt<10>(); break; case 11: wait_vmcnt<11>(); break;
                case 12: wait_vmcnt<12>(); break; case 13: wait_vmcnt<13>(); break; case 14: wait_vmcnt<14>(); break; case 15: wait_vmcnt<15>(); break;
                case 16: wait_vmcnt<16>(); break; case 17: wait_vmcnt<17>(); break; case 18: wait_vmcnt<18>(); break; case 19: wait_vmcnt<19>(); break;
                case 20: wait_vmcnt<20>(); break; case 21: wait_vmcnt<21>(); break; case 22: wait_vmcnt<22>(); break; case 23: wait_vmcnt<23>(); break;
                case 24: wait_vmcnt<24>(); break; case 25: wait_vmcnt<25>(); break; case 26: wait_vmcnt<26>(); break; case 27: wait_vmcnt<27>(); break;
                default: wait_vmcnt<28>(); break;   // (a smaller count than allowed is only stricter)
            }
        };
        const int win_pieces = wrows * CPR;                    // per plane
        const int win_total = win_pieces * NPL;
        u32x4 v[WV];
        // Window of a chunk -> buffer chunk & 1, in two halves that sit a chunk of k-steps apart: the global loads go out right behind the
        // barrier of the chunk's first tap (into registers), the LDS writes behind the barrier of its last tap -- the loaders never sit
        // in a global-load latency between two barriers.
        auto fetch_window = [&](int chunk) {
#pragma unroll
            for (int i = 0; i < WV; i++) {
                const int idx = min(ll + 256 * i, win_total - 1);
                const int pl = idx / win_pieces, rem = idx - pl * win_pieces;
                const int w = rem / CPR, c = rem - w * CPR;
                const int g = m0 - halo + w;
                const int gc = min(max(g, sstart), send - 1);   // always a valid row; rows outside the sequence are zeroed at the LDS write
                v[i] = *reinterpret_cast<const u32x4*>(p.A[NPL == 2 ? pl : 0] + (size_t)gc * p.lda + a_col0 + (size_t)chunk * CKC + c * 8);
            }
        };
        auto store_window = [&](int chunk) {
            char* dstb = win0 + (chunk & wsel) * win_buf;
#pragma unroll
            for (int i = 0; i < WV; i++) {
                const int idx = ll + 256 * i;
                const int pl = idx / win_pieces, rem = idx - pl * win_pieces;
                const int w = rem / CPR, c = rem - w * CPR;
                const int g = m0 - halo + w;
                const bool ok = g >= sstart && g < send;          // (the select sits here, not behind the load: the data is first touched now)
                if (idx < win_total) *reinterpret_cast<u32x4*>(dstb + pl * win_plane + c5_off<ROWB>(w, c)) = ok ? v[i] : (u32x4){0u, 0u, 0u, 0u};
            }
        };
#pragma unroll
        for (int s = 0; s < NST; s++)
            if (s < nk) issue_w(s);
        fetch_window(0);
        store_window(0);
        if constexpr (ABL) ts[1] = C5_NOW();
        int win_age = NST;                                       // barriers since the last fetch; the fetch follows the issue of W(kt - 1 + NST), so it is
                                                                 // younger than the group waited for at B_{kt + j} exactly while j <= NST - 1
        for (int kt = 0; kt < nk; kt++) {
            const int newer = kt == 0 ? min(NST - 1, nk - 1) : min(NST - 2, nk - 1 - kt);
            unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
            if constexpr (ABL) t0 = C5_NOW();
            wait_outstanding(newer * mine + (win_age <= NST - 1 ? WV : 0));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the window writes of this wave are in LDS
            if constexpr (ABL) t1 = C5_NOW();
            __builtin_amdgcn_s_barrier();                        // B_kt
            if constexpr (ABL) t2 = C5_NOW();
            const int chunk = kt / taps, tap = kt - chunk * taps;
            const bool more = chunk + 1 < nchunks;
            if constexpr (SINGLE_WIN) {
                // behind B_kt of a chunk's FIRST tap every consumer is done with the chunk before: write this chunk's window (fetched a chunk
                // ago) over it, then a second barrier releases the consumers
                if (tap == 0 && chunk > 0) {
                    store_window(chunk);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                // B'_kt
                }
            } else {
                if (more && tap == taps - 1) store_window(chunk + 1);   // BEFORE the next issue: the compiler's wait for v[] then covers only older groups
            }
            if constexpr (ABL) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t3 = C5_NOW(); }
            if (kt >= 1 && kt - 1 + NST < nk) issue_w(kt - 1 + NST);
            if (more && tap == 0) {
                fetch_window(chunk + 1);
                win_age = 0;
            }
            win_age++;
            if constexpr (ABL) { t4 = C5_NOW(); ts[3] += t1 - t0; ts[4] += t2 - t1; ts[5] += t3 - t2; ts[6] += t4 - t3; }
        }
        if constexpr (ABL) {
            ts[2] = C5_NOW();
            if (p.stamps && tid == 256) {
                unsigned long long* o = p.stamps + (size_t)blockIdx.x * 16 + 8;
                for (int i = 0; i < 7; i++) o[i] = ts[i];
            }
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------------- consumer waves
    // Software pipeline over HALF k-steps (HK = KS / 2 MFMA k-substeps each): two register sets of fragments.  While the MFMAs of one
    // half issue, the reads of the other are in flight; barrier B_{kt+1} sits in the MIDDLE of k-step kt, behind the return of all its
    // reads (so the loaders may refill its stage) and in front of the prefetch of k-step kt + 1's first half -- no MFMA ever waits on a
    // read that was issued behind a barrier it also waited on.
    const int fr = lane & 31, fh = lane >> 5;
    constexpr int HK = KS / 2;
    static_assert(KS == 2 || KS == 4, "one or two MFMA k-substeps per half k-step");
    struct Half { bf16x8 a[HK][NPL][RBW], b[HK][NPL][NB]; };
    f32x16 acc[RBW][NB];
#pragma unroll
    for (int i = 0; i < RBW; i++)
#pragma unroll
        for (int j = 0; j < NB; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc[i][j][g] = 0.0f;
    const int center = p.conv_center;
    const int row_base = wave * (32 * RBW) + halo + fr;          // window row of this lane's first A-fragment row at tap == center
    // fragments of half `h` of the k-step (chunk, tap) whose weight tile sits in stage `stage`
    auto load_half = [&](Half& f, int chunk, int tap, int stage, int h) {
        const char* win = win0 + (chunk & wsel) * win_buf;
        const char* wst = wst0 + stage * WST;
        const int wrow = row_base + (tap - center) * dil;
#pragma unroll
        for (int q = 0; q < HK; q++) {
            const int c16 = (h * HK + q) * 2 + fh;
#pragma unroll
            for (int pl = 0; pl < NPL; pl++) {
#pragma unroll
                for (int i = 0; i < RBW; i++) f.a[q][pl][i] = *reinterpret_cast<const bf16x8*>(win + pl * win_plane + c5_off<ROWB>(wrow + i * 32, c16));
#pragma unroll
                for (int j = 0; j < NB; j++) f.b[q][pl][j] = *reinterpret_cast<const bf16x8*>(wst + pl * (BN * ROWB) + c5_off<ROWB>(j * 32 + fr, c16));
            }
        }
    };
    auto mfma_half = [&](const Half& f) {
#pragma unroll
        for (int q = 0; q < HK; q++)
#pragma unroll
            for (int i = 0; i < RBW; i++)
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    if constexpr (NPL == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[q][1][i], f.b[q][0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[q][0][i], f.b[q][1][j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = mfma_32x32x16<F16>(f.a[q][0][i], f.b[q][0][j], acc[i][j]);
                }
    };
    int chunk = 0, tap = 0, stage = 0;                           // of k-step kt (wave-uniform counters: no division in the loop)
    if constexpr (NB <= 2) {
        // Narrow tiles (24-64 output channels, 3-33 k-steps): one fragment set, barrier at the top of the k-step.  They fit 128
        // registers, so two workgroups share a CU and the second one covers this one's window fetch, barriers and epilogue.
        Half h;
        for (int kt = 0; kt < nk; kt++) {
            __builtin_amdgcn_s_barrier();                        // B_kt
            if constexpr (SINGLE_WIN) {
                if (tap == 0 && chunk > 0) __builtin_amdgcn_s_barrier();   // B'_kt: the loaders have rewritten the (single) window buffer
            }
            load_half(h, chunk, tap, stage, 0);
            mfma_half(h);
            load_half(h, chunk, tap, stage, 1);
            mfma_half(h);
            if (++tap == taps) { tap = 0; chunk++; }
            if (++stage == NST) stage = 0;
        }
    } else {
        Half h0, h1;
        __builtin_amdgcn_s_barrier();                                // B_0
        if constexpr (ABL) ts[1] = C5_NOW();
        load_half(h0, 0, 0, 0, 0);
        // (the last k-step is peeled: with the barrier under a condition the two paths merge in front of the second half's MFMAs, and the
        // compiler then makes them wait on the prefetch reads of the barrier path)
        // Issue order inside a half k-step, pinned with sched_group_barrier: one fragment read of the OTHER half behind each of the first
        // MFMAs of this half.  Left alone the scheduler, short of registers, sinks every read back to just in front of its first use (the
        // MFMAs then wait on LDS latency); all reads in one block in front of the MFMAs costs ~150 cycles per half in which no MFMA issues.
        constexpr int READS = HK * NPL * (RBW + NB), MFMAS = HK * RBW * NB * (NPL == 2 ? 3 : 1);
        constexpr int PAIRS = READS < MFMAS ? READS : MFMAS;
        auto interleave = [&]() {
    #pragma unroll
            for (int r = 0; r < PAIRS; r++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one DS read
            }
            if constexpr (MFMAS > PAIRS) __builtin_amdgcn_sched_group_barrier(0x008, MFMAS - PAIRS, 0);
            if constexpr (READS > PAIRS) __builtin_amdgcn_sched_group_barrier(0x100, READS - PAIRS, 0);
        };
        for (int kt = 0; kt + 1 < nk; kt++) {
            load_half(h1, chunk, tap, stage, 1);
            mfma_half(h0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            if (++tap == taps) { tap = 0; chunk++; }
            if (++stage == NST) stage = 0;
            __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0) (the builtin: the compiler then knows every read has returned)
            unsigned long long tb = 0;
            if constexpr (ABL) tb = C5_NOW();
            __builtin_amdgcn_s_barrier();                            // B_{kt+1}
            if constexpr (ABL) ts[3] += C5_NOW() - tb;
            __builtin_amdgcn_sched_barrier(0);
            load_half(h0, chunk, tap, stage, 0);
            mfma_half(h1);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
        }
        load_half(h1, chunk, tap, stage, 1);
        mfma_half(h0);
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(h1);
    }
    if constexpr (ABL) ts[2] = C5_NOW();

    // ---------------------------------------------------------------------------------------------------- epilogue
    // accumulator g of block (i, j): row = i * 32 + (g & 3) + 8 * (g >> 2) + 4 * fh, column = j * 32 + fr.
    // BigVGAN tiles (RBW == 2): v = acc + bias + res -> fp32 rows, nothing else compiled in (these kernels sit at ~240 registers).
    // General epilogue (GEN: the 128-row grouped variant): v = act(acc + bias) + res -> fp32 rows and / or split-bf16 planes; grouped
    // convolution (p.group_w > 0: the column tile is a group padded to BN columns): real column = group * group_w + local column,
    // local columns >= group_w are padding.
    constexpr bool GEN = RBW == 1;
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const int lc = j * 32 + fr;
        const int col = GEN && p.group_w > 0 ? tn * p.group_w + lc : n0 + lc;
        if (GEN && p.group_w > 0 ? lc >= p.group_w : col >= p.N) continue;
        const float b = p.bias ? p.bias[n0 + lc] : 0.0f;
#pragma unroll
        for (int i = 0; i < RBW; i++) {
            const int rbase = m0 + wave * (32 * RBW) + i * 32 + 4 * fh;
            // residual loads in groups of GR rows: all 16 at once on the wide tiles (one workgroup per CU: memory-level parallelism),
            // 4 at a time on the narrow ones (128-register budget; the second workgroup on the CU supplies the overlap)
            constexpr int GR = NB <= 2 ? 4 : 16;
#pragma unroll
            for (int g0 = 0; g0 < 16; g0 += GR) {
                float r[GR];
                if (p.res) {
#pragma unroll
                    for (int g = 0; g < GR; g++) r[g] = p.res[(size_t)(rbase + ((g0 + g) & 3) + 8 * ((g0 + g) >> 2)) * p.ldres + col];
                }
#pragma unroll
                for (int g = 0; g < GR; g++) {
                    const size_t row = (size_t)(rbase + ((g0 + g) & 3) + 8 * ((g0 + g) >> 2));
                    float v = acc[i][j][g0 + g] + b;
                    if constexpr (GEN) {
                        if (p.act == ACT_MISH) v = apply_act(v, ACT_MISH);
                    }
                    if (p.res) v += r[g];
                    if constexpr (GEN) {
                        if (p.out_f32) p.out_f32[row * p.ldo + col] = v;
                        if (p.out_hi) {
                            __bf16 hi, lo;
                            split_bf16(v, hi, lo);
                            p.out_hi[row * p.ldob + col] = hi;
                            if (p.out_lo) p.out_lo[row * p.ldob + col] = lo;
                        }
                    } else {
                        p.out_f32[row * p.ldo + col] = v;
                    }
                }
            }
        }
    }
    if constexpr (ABL) {
        if (p.stamps && tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ts[4] = C5_NOW();
            unsigned long long* o = p.stamps + (size_t)blockIdx.x * 16;
            for (int i = 0; i < 5; i++) o[i] = ts[i];
        }
    }
#undef C5_NOW
}

template <int NPL, int ROWB, int NB, bool F16, int ABL = 0, int RBW = 2>
static hipError_t launch_conv5_t(const GemmArgs& a, int n_pad, int taps, hipStream_t st) {
    using C = Conv5Cfg<NPL, ROWB, NB, RBW>;
    const int dil = a.conv_dil > 1 ? a.conv_dil : 1;
    const int halo = a.conv_center * dil;
    const int nchunks = (a.K / taps) / C::CKC;
    const int lds = C::lds_bytes(halo, nchunks);
    static unsigned attr_mask = 0;
    if (hipError_t e = f5_set_lds_attr(reinterpret_cast<const void*>(&conv5_kernel<NPL, ROWB, NB, F16, ABL, RBW>), 160 * 1024, attr_mask); e != hipSuccess) return e;
    const int tiles_m = a.M / C::BM, tiles_n = n_pad / C::BN;
    hipLaunchKernelGGL((conv5_kernel<NPL, ROWB, NB, F16, ABL, RBW>), dim3(tiles_m * tiles_n), dim3(512), lds, st, a, tiles_m, taps, halo, C::win_rows(halo));
    return hipGetLastError();
}

// prec: 1 = bf16, 2 = split bf16, 3 = fp16 (one plane).  hipErrorInvalidValue: shape not covered (the caller falls back to gemm.h).
static hipError_t launch_conv5(int prec, const GemmArgs& a, int n_pad, hipStream_t st) {
    if (a.conv_group_cols > 0) {
        // grouped Conv1d of the DiT's ConvPositionEmbedding (k = 31, 16 groups padded to 64 columns / 64 channels each, split bf16, Mish,
        // optional fp32 residual, fp32 rows or split-bf16 planes out): 128-row tiles, one column tile per group.  The sequence bounds
        // come from the tile's first row, so every sequence must start at a multiple of 128 rows (the packed layout pads to that).
        if (prec != 2 || a.group_w != a.conv_group_cols || a.group_w > 64 || n_pad % 64 || a.mul || a.row_keep || !a.row_seq_start || !a.row_seq_end ||
            (a.act != ACT_NONE && a.act != ACT_MISH) ||
            (!a.out_f32 && !a.out_hi) || a.M % 128 || a.conv_center <= 0 || a.conv_center > 25 || (a.conv_dil > 1))
            return hipErrorInvalidValue;
        const int taps = 2 * a.conv_center + 1;
        if (a.K % taps || a.K / taps != 64) return hipErrorInvalidValue;
        return launch_conv5_t<2, 64, 2, false, 0, 1>(a, n_pad, taps, st);
    }
    if (a.row_seq_start || a.group_w || a.act || a.mul || a.row_keep || a.out_hi || !a.out_f32) return hipErrorInvalidValue;
    if (a.lda <= 0 || a.K % a.lda) return hipErrorInvalidValue;
    const int taps = a.K / a.lda;
    const int dil = a.conv_dil > 1 ? a.conv_dil : 1;
    if (taps != 2 * a.conv_center + 1 || taps < 3 || a.conv_center * dil > 25) return hipErrorInvalidValue;
    if (a.M % 256 || a.seq_pitch <= 0 || a.seq_pitch % 256) return hipErrorInvalidValue;
    const bool wide = n_pad > 128;
    if (wide ? (n_pad % 128 != 0) : (n_pad != 64 && n_pad != 128)) return hipErrorInvalidValue;
    const int nb = wide ? 4 : n_pad / 32;
    if (a.stamps) {   // diagnostics build of the two wide configurations
        if (prec == 2 && nb == 4 && a.lda % 32 == 0) return launch_conv5_t<2, 64, 4, false, 1>(a, n_pad, taps, st);
        if (prec == 3 && nb == 4 && a.lda % 64 == 0) return launch_conv5_t<1, 128, 4, true, 1>(a, n_pad, taps, st);
        return hipErrorInvalidValue;
    }
    if (prec == 2) {
        if (a.lda % 32) return hipErrorInvalidValue;
        return nb == 4 ? launch_conv5_t<2, 64, 4, false>(a, n_pad, taps, st) : launch_conv5_t<2, 64, 2, false>(a, n_pad, taps, st);
    }
    if (prec == 3) {
        if (a.lda % 64 == 0) return nb == 4 ? launch_conv5_t<1, 128, 4, true>(a, n_pad, taps, st) : launch_conv5_t<1, 128, 2, true>(a, n_pad, taps, st);
        if (a.lda % 32) return hipErrorInvalidValue;
        return nb == 4 ? launch_conv5_t<1, 64, 4, true>(a, n_pad, taps, st) : launch_conv5_t<1, 64, 2, true>(a, n_pad, taps, st);
    }
    return hipErrorInvalidValue;
}
