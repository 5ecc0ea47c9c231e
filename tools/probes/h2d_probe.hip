// Host -> device copy rates on the GPU box: pageable hipMemcpy (what fit_host did), pinned hipMemcpyAsync (the PCIe ceiling), and
// pageable memory staged through pinned slots by T host threads, each with its own stream (the design of amx_stage.hpp).
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -o /tmp/h2d_probe tools/probes/h2d_probe.hip -lpthread && /tmp/h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void staged(char *dst, const char *src, size_t bytes, int T, size_t slice, std::vector<char *> &pin, std::vector<hipStream_t> &st, std::vector<hipEvent_t> &ev)
{
    const size_t n_sl = (bytes + slice - 1) / slice;
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            CK(hipSetDevice(0));
            int k = 0;
            for (size_t i = t; i < n_sl; i += T, k++) {
                const int slot = 2 * t + (k & 1);
                if (k >= 2) CK(hipEventSynchronize(ev[slot]));
                const size_t o = i * slice, n = std::min(slice, bytes - o);
                memcpy(pin[slot], src + o, n);
                CK(hipMemcpyAsync(dst + o, pin[slot], n, hipMemcpyHostToDevice, st[t]));
                CK(hipEventRecord(ev[slot], st[t]));
            }
            CK(hipStreamSynchronize(st[t]));
        });
    for (auto &x : th) x.join();
}

#include <immintrin.h>
#include <atomic>
__attribute__((target("avx2"))) static bool narrow_intr(const double *__restrict__ s, float *__restrict__ d, size_t n)
{
    __m256d acc = _mm256_setzero_pd();
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256d a = _mm256_loadu_pd(s + i), b = _mm256_loadu_pd(s + i + 4);
        const __m128 fa = _mm256_cvtpd_ps(a), fb = _mm256_cvtpd_ps(b);
        _mm256_storeu_ps(d + i, _mm256_set_m128(fb, fa));
        acc = _mm256_or_pd(acc, _mm256_cmp_pd(_mm256_cvtps_pd(fa), a, _CMP_NEQ_UQ));
        acc = _mm256_or_pd(acc, _mm256_cmp_pd(_mm256_cvtps_pd(fb), b, _CMP_NEQ_UQ));
    }
    bool bad = _mm256_movemask_pd(acc) != 0;
    for (; i < n; i++) { const float f = (float)s[i]; d[i] = f; bad = bad || ((double)f != s[i]); }
    return !bad;
}
// mode 0: narrow only; 1: narrow + send on the thread's own stream; 2: narrow, one sender thread (this one) sends finished slices
static void narrowed(float *dst, const double *src, size_t n_el, int T, size_t slice_el, int mode, std::vector<float *> &pin, std::vector<hipStream_t> &st, std::vector<hipEvent_t> &ev)
{
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            CK(hipSetDevice(0));
            bool used[2] = {false, false};
            for (int k = 0;; k++) {
                const size_t o = next.fetch_add(1) * slice_el;
                if (o >= n_el) break;
                const size_t n = std::min(slice_el, n_el - o);
                const int slot = 2 * t + (k & 1);
                if (mode == 1 && used[k & 1]) CK(hipEventSynchronize(ev[slot]));
                narrow_intr(src + o, pin[slot], n);
                if (mode == 1) { CK(hipMemcpyAsync(dst + o, pin[slot], n * 4, hipMemcpyHostToDevice, st[t])); CK(hipEventRecord(ev[slot], st[t])); used[k & 1] = true; }
            }
            if (mode == 1) CK(hipStreamSynchronize(st[t]));
        });
    for (auto &x : th) x.join();
}

int main()
{
    const size_t bytes = 792ull << 20;
    char *src = (char *)malloc(bytes);
    for (size_t i = 0; i < bytes; i += 4096) src[i] = (char)i;          // touch
    memset(src, 1, bytes);
    char *dst; CK(hipMalloc(&dst, bytes));
    char *pinned; CK(hipHostMalloc(&pinned, bytes)); memset(pinned, 2, bytes);
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now(); CK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); double t1 = now();
        printf("pageable hipMemcpy          %6.2f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now(); CK(hipMemcpy(dst, pinned, bytes, hipMemcpyHostToDevice)); double t1 = now();
        printf("pinned hipMemcpy            %6.2f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
    {   // host memcpy rate of one thread, pageable -> pinned
        double t0 = now(); memcpy(pinned, src, bytes); double t1 = now();
        printf("one-thread memcpy           %6.2f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
    for (int T : {4, 8})
        for (size_t slice : {size_t(2) << 20}) {
            std::vector<char *> pin(2 * T); std::vector<hipStream_t> st(T); std::vector<hipEvent_t> ev(2 * T);
            for (auto &p : pin) CK(hipHostMalloc(&p, slice));
            for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            double best = 1e9;
            for (int rep = 0; rep < 4; rep++) { double t0 = now(); staged(dst, src, bytes, T, slice, pin, st, ev); double t1 = now(); best = std::min(best, t1 - t0); }
            printf("staged T=%2d slice=%zu MB      %6.2f ms  %5.1f GB/s\n", T, slice >> 20, best * 1e3, bytes / best / 1e9);
            for (auto &p : pin) CK(hipHostFree(p));
            for (auto &s : st) CK(hipStreamDestroy(s));
            for (auto &e : ev) CK(hipEventDestroy(e));
        }
    {
        const size_t n_el = bytes / 8;
        double *sd = (double *)src;
        for (size_t i = 0; i < n_el; i++) sd[i] = (double)(float)(1.0 + 1e-7 * (double)(i & 0xfffff));
        for (int mode : {0, 1})
            for (int T : {4, 8, 12, 16, 32}) {
                const size_t slice_el = 512 * 1024;
                std::vector<float *> pin(2 * T); std::vector<hipStream_t> st(T); std::vector<hipEvent_t> ev(2 * T);
                for (auto &p : pin) CK(hipHostMalloc(&p, slice_el * 4));
                for (auto &q : st) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
                for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                double best = 1e9, worst = 0;
                for (int rep = 0; rep < 5; rep++) { double t0 = now(); narrowed((float *)dst, sd, n_el, T, slice_el, mode, pin, st, ev); double t1 = now(); best = std::min(best, t1 - t0); worst = std::max(worst, t1 - t0); }
                printf("narrow %s T=%2d   best %6.2f ms (%5.1f GB/s of float64)  worst %6.2f ms\n", mode ? "+ send" : "only  ", T, best * 1e3, bytes / best / 1e9, worst * 1e3);
                for (auto &p : pin) CK(hipHostFree(p));
                for (auto &q : st) CK(hipStreamDestroy(q));
                for (auto &e : ev) CK(hipEventDestroy(e));
            }
    }
    // a smaller copy (one batch of the float32 path: 393 216 voxels x 99 x 4 B = 156 MB)
    const size_t b2 = 156ull << 20;
    for (int rep = 0; rep < 2; rep++) { double t0 = now(); CK(hipMemcpy(dst, src, b2, hipMemcpyHostToDevice)); double t1 = now(); printf("pageable 156 MB             %6.2f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, b2 / (t1 - t0) / 1e9); }
    // device -> host, pageable vs pinned (the maps: 24 .. 48 MB)
    const size_t b3 = 48ull << 20;
    for (int rep = 0; rep < 2; rep++) { double t0 = now(); CK(hipMemcpy(src, dst, b3, hipMemcpyDeviceToHost)); double t1 = now(); printf("D2H pageable 48 MB          %6.2f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, b3 / (t1 - t0) / 1e9); }
    for (int rep = 0; rep < 2; rep++) { double t0 = now(); CK(hipMemcpy(pinned, dst, b3, hipMemcpyDeviceToHost)); double t1 = now(); printf("D2H pinned 48 MB            %6.2f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, b3 / (t1 - t0) / 1e9); }
    return 0;
}
