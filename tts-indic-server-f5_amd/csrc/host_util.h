// Host-side plumbing of libf5hip: error reporting, device allocations, weight packing, HIP-event profiling.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include <string.h>

#include "common.h"
#include "elementwise.h"

// ---------------------------------------------------------------- errors
static thread_local char g_err[512] = "";
static void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char* f5hip_last_error(void) { return g_err; }
extern "C" int f5hip_abi_version(void) { return 1; }

// ---------------------------------------------------------------- device memory
struct Plane2 { __bf16* hi = nullptr; __bf16* lo = nullptr; };
struct DevBuf { void* ptr = nullptr; };
struct PackedW {
    __bf16* hi = nullptr; __bf16* lo = nullptr; float* bias = nullptr;
    __bf16* frag = nullptr;   // f16 weights once more in MFMA-fragment order (pack_frag: the W-direct gemm5 kernels), or null
    int n = 0, k = 0, n_pad = 0, k_pad = 0, ld = 0;
    bool f16 = false;   // hi holds one fp16 plane, lo == nullptr
};

template <typename T> static void dev_free(T* p) { if (p) (void)hipFree((void*)p); }

// bump allocator over one hipMalloc'ed arena (256-byte aligned carves); base == nullptr only measures
struct Arena {
    char* base = nullptr; size_t off = 0;
    void reset(char* b) { base = b; off = 0; }
    void* take(size_t bytes) {
        void* p = base ? base + off : nullptr;
        off += (bytes + 255) / 256 * 256;
        return p;
    }
    float* f32(size_t n) { return (float*)take(n * 4); }
    __bf16* bf16(size_t n) { return (__bf16*)take(n * 2); }
    Plane2 plane2(size_t n) { Plane2 p; p.hi = bf16(n); p.lo = bf16(n); return p; }
    size_t used() const { return off; }
};

static int upload_f32(float** dst, const float* src, size_t n) {
    if (hipMalloc((void**)dst, n * sizeof(float)) != hipSuccess) { *dst = nullptr; return fail(-4, "hipMalloc %zu floats", n); }
    if (hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail(-4, "hipMemcpy H2D");
    return 0;
}

// round-to-nearest-even fp32 -> bf16 on the host (finite inputs)
static uint16_t host_bf16(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float host_bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static void host_split_bf16(float x, uint16_t& hi, uint16_t& lo) {
    hi = host_bf16(x);
    lo = host_bf16(x - host_bf16_to_f32(hi));
}

// W [N][K] fp32 host (row stride ldw) -> split bf16 device [ceil128(N)][ceil32(K)], bias -> fp32 [ceil128(N)]
static int pack_linear(PackedW& out, const float* w, int N, int K, int ldw, const float* bias, int n_align = 128, bool f16 = false) {
    out.f16 = f16;
    out.n = N; out.k = K; out.n_pad = (N + n_align - 1) / n_align * n_align; out.k_pad = (K + 31) / 32 * 32; out.ld = out.k_pad;
    const size_t np = (size_t)out.n_pad * out.k_pad;
    float* tmp = nullptr;
    if (hipMalloc((void**)&tmp, (size_t)N * ldw * sizeof(float)) != hipSuccess) return fail(-4, "hipMalloc pack staging");
    if (hipMalloc((void**)&out.hi, np * 2) != hipSuccess || (!f16 && hipMalloc((void**)&out.lo, np * 2) != hipSuccess)) {
        dev_free(tmp);
        return fail(-4, "hipMalloc packed weight %d x %d", out.n_pad, out.k_pad);
    }
    if (hipMemcpy(tmp, w, (size_t)N * ldw * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { dev_free(tmp); return fail(-4, "H2D weight"); }
    hipLaunchKernelGGL(pack_weight_kernel, dim3(out.n_pad), dim3(256), 0, 0, tmp, N, K, ldw, out.hi, out.lo, out.k_pad);
    if (hipDeviceSynchronize() != hipSuccess) { dev_free(tmp); return fail(-4, "pack_weight_kernel"); }
    dev_free(tmp);
    std::vector<float> b(out.n_pad, 0.0f);
    if (bias) std::copy(bias, bias + N, b.begin());
    return upload_f32(&out.bias, b.data(), b.size());
}

// Per-call host -> device uploads (row metadata, the sinusoid table) without a host sync: the bytes go through a pinned staging buffer owned by
// the handle, the copy is queued on the caller's stream, and an event recorded behind it guards the buffer's NEXT use (already complete by then
// in practice).  The call returns with nothing but stream-ordered work outstanding, so a caller may capture it or keep several streams busy.
struct HostStage {
    void* ptr = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false;
    int upload(void* dst_dev, const void* src, size_t bytes, hipStream_t st) {
        if (pending) { (void)hipEventSynchronize(ev); pending = false; }
        if (bytes > cap) {
            if (ptr) (void)hipHostFree(ptr);
            ptr = nullptr; cap = 0;
            const size_t want = (bytes + 65535) / 65536 * 65536;
            if (hipHostMalloc(&ptr, want, hipHostMallocDefault) != hipSuccess) { ptr = nullptr; return fail(-5, "hipHostMalloc %zu bytes of staging", want); }
            cap = want;
        }
        if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ev = nullptr; return fail(-5, "hipEventCreate"); }
        memcpy(ptr, src, bytes);
        if (hipMemcpyAsync(dst_dev, ptr, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return fail(-6, "staged upload");
        if (hipEventRecord(ev, st) != hipSuccess) return fail(-6, "staged upload: event");
        pending = true;
        return 0;
    }
    void release() {
        if (pending) (void)hipEventSynchronize(ev);
        if (ev) (void)hipEventDestroy(ev);
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr; ev = nullptr; cap = 0; pending = false;
    }
};

// fragment-ordered second copy of a one-plane fp16 weight (K a multiple of 128, rows a multiple of 16): GemmArgs::Wf
static int pack_frag(PackedW& w, hipStream_t st = 0) {
    if (!w.f16 || !w.hi || w.k_pad % 128 || w.ld != w.k_pad || w.n_pad % 16 || w.frag) return 0;
    const size_t n = (size_t)w.n_pad * w.k_pad;
    if (hipMalloc((void**)&w.frag, n * 2) != hipSuccess) { w.frag = nullptr; return fail(-4, "hipMalloc fragment-ordered weight %d x %d", w.n_pad, w.k_pad); }
    hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, st, w.hi, w.n_pad, w.k_pad, w.ld, w.frag);
    return hipGetLastError() == hipSuccess ? 0 : fail(-4, "pack_frag_kernel");
}

// ---------------------------------------------------------------- HIP-event profiling per kernel class
enum { PROF_GEMM = 0, PROF_ATTN = 1, PROF_LN = 2, PROF_OTHER = 3, PROF_VOCOS = 4, PROF_N = 5 };
static const char* g_prof_names[PROF_N] = {"gemm", "attn", "ln", "other", "vocos"};
struct ProfSpan { int cls; hipEvent_t a, b; };
// One profiling state per owner: the process-wide default (f5hip_set_profiling / f5hip_get_profile) and one per sampler handle that asked for
// its own (f5hip_dit_set_profiling).  The ABI entry points select the state of their handle for the calling THREAD (ProfScope), so two
// handles driven from two threads never share spans, pools or totals.
struct ProfState {
    std::mutex mu;   // the process-wide state can be reached from several threads (handles without a state of their own)
    bool on = false;
    std::vector<ProfSpan> spans;
    std::vector<hipEvent_t> pool;
    double ms[PROF_N] = {0, 0, 0, 0, 0};
    long long cnt[PROF_N] = {0, 0, 0, 0, 0};
    int open = 0;
    void collect() {
        for (auto& s : spans) {
            float t = 0.0f;
            (void)hipEventSynchronize(s.b);
            if (hipEventElapsedTime(&t, s.a, s.b) == hipSuccess) { ms[s.cls] += t; cnt[s.cls]++; }
            pool.push_back(s.a); pool.push_back(s.b);
        }
        spans.clear();
    }
    void set(bool enabled) {
        std::lock_guard<std::mutex> lk(mu);
        collect();
        on = enabled;
        for (int i = 0; i < PROF_N; i++) { ms[i] = 0.0; cnt[i] = 0; }
    }
    int get(const char* kernel_class, double* total_ms, int64_t* launches) {
        std::lock_guard<std::mutex> lk(mu);
        collect();
        for (int i = 0; i < PROF_N; i++)
            if (!strcmp(kernel_class, g_prof_names[i])) {
                if (total_ms) *total_ms = ms[i];
                if (launches) *launches = cnt[i];
                return 0;
            }
        return fail(-1, "unknown kernel class %s", kernel_class);
    }
    ~ProfState() { for (hipEvent_t e : pool) (void)hipEventDestroy(e); }
};
static ProfState g_prof_default;
static thread_local ProfState* t_prof = &g_prof_default;
struct ProfScope {   // selects `p` (a handle's own state; null = leave the thread on what it has) for the current thread until the scope ends
    ProfState* prev;
    explicit ProfScope(ProfState* p) : prev(t_prof) { if (p) t_prof = p; }
    ~ProfScope() { t_prof = prev; }
};

static hipEvent_t prof_event(ProfState& ps) {
    if (!ps.pool.empty()) { hipEvent_t e = ps.pool.back(); ps.pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
static void prof_begin(int cls, hipStream_t st) {
    ProfState& ps = *t_prof;
    if (!ps.on) return;
    std::lock_guard<std::mutex> lk(ps.mu);
    if (ps.open++) return;   // nested spans are attributed to the outer class
    ProfSpan s; s.cls = cls; s.a = prof_event(ps); s.b = prof_event(ps);
    (void)hipEventRecord(s.a, st);
    ps.spans.push_back(s);
}
static void prof_end(int cls, hipStream_t st) {
    (void)cls;
    ProfState& ps = *t_prof;
    if (!ps.on) return;
    std::lock_guard<std::mutex> lk(ps.mu);
    if (--ps.open) return;
    (void)hipEventRecord(ps.spans.back().b, st);
}
extern "C" int f5hip_set_profiling(int32_t enabled) {
    g_prof_default.set(enabled != 0);
    return 0;
}
extern "C" int f5hip_get_profile(const char* kernel_class, double* total_ms, int64_t* launches) {
    return g_prof_default.get(kernel_class, total_ms, launches);
}
