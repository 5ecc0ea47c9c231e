// amx_stage.hpp -- float64 signals in pageable host memory travel to the device as float32 when that loses nothing.
//
// The reference's boundary is host numpy in, host numpy out (core.py:465-466), and evaluation.y is the float64 CAST of a
// float32 image (core.py:136 loads niiDWI_img as float32, core.py:209-223 normalises it in place, models.pyx:902 reads the
// float64 copy): every value of y is exactly a float32.  A host-buffer fit of 1 M NODDI voxels is 14.9 ms of PCIe
// (792 MB at the link's 56.4 GB/s, measured per batch inside the call: AMX_HOST_TRACE=1, profiles/r05c_host_trace.txt) next
// to 6.5 ms of kernels -- the copy IS the call.  So the copy is halved where that is exact: host threads narrow the caller's
// buffer chunk by chunk into two pinned buffers, CHECKING every element ((double)(float)v == v; 144 - 260 GB/s of float64 on
// the box's cores, tools/probes/h2d_probe.hip), the calling thread sends chunk k with the blocking copy it always used while
// chunk k + 1 is being narrowed, and the device widens the batch again (k_widen, amx_api.hip).  One element that is not a
// float32 (or is a NaN: it never compares equal) and the batch -- and every later batch of the call -- is copied as it is.
// The values the kernels read are the caller's, bit for bit, either way.
// (Measured and dropped: every host thread sending its own slices on its own stream -- 100 GB/s of float64 alone on the box,
//  30 - 80 GB/s while the solver runs, from one batch to the next: profiles/r05c_host_trace.txt.)
#pragma once
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>
#include <cstdio>
#include <pthread.h>
#include <sched.h>

namespace amx_stage {

// The CPUs of one NUMA node (Linux sysfs), intersected with the CPUs this process may use.  false: unknown -- leave the threads alone.
static bool node_cpus(int node, cpu_set_t *out)
{
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) { fclose(f); return false; }
    CPU_ZERO(out);
    int a, b, n = 0;
    for (;;) {
        if (fscanf(f, "%d", &a) != 1) break;
        b = a;
        int c = fgetc(f);
        if (c == '-') { if (fscanf(f, "%d", &b) != 1) break; c = fgetc(f); }
        for (int k = a; k <= b && k < CPU_SETSIZE; k++) if (CPU_ISSET(k, &allowed)) { CPU_SET(k, out); n++; }
        if (c != ',') break;
    }
    fclose(f);
    return n > 0;
}
// NUMA node a HIP device hangs on (-1: unknown / single node)
static int device_node(int device)
{
    char bus[64] = {0}, path[160];
    if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *p = bus; *p; p++) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

// true: every element narrowed exactly (a NaN compares unequal -> false)
__attribute__((target("avx2"))) static bool narrow_avx2(const double *__restrict__ s, float *__restrict__ d, size_t n)
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
static bool narrow_base(const double *__restrict__ s, float *__restrict__ d, size_t n)
{
    unsigned bad = 0u;
    for (size_t i = 0; i < n; i++) { const float f = (float)s[i]; d[i] = f; bad |= (unsigned)((double)f != s[i]); }
    return bad == 0u;
}

// T host threads that narrow one chunk at a time (start / wait: one job in flight)
class Pool {
public:
    static constexpr size_t kChunkEl = 4u << 20;           // elements per chunk: 32 MB read, 16 MB sent (0.28 ms of link)
    static constexpr size_t kPieceEl = 32u << 10;          // elements a thread takes at a time
    // node >= 0: the threads stay on that NUMA node's CPUs
    static Pool *create(int threads, int node = -1)
    {
        Pool *p = new Pool();
        p->T_ = threads;
        p->avx2_ = __builtin_cpu_supports("avx2");
        if (const char *e = getenv("AMX_HOST_SPIN_US")) p->spin_us_ = atoi(e);
        for (float *&q : p->ring_)
            if (hipHostMalloc((void **)&q, kChunkEl * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); q = nullptr; p->release(); delete p; return nullptr; }
        try {
            for (int t = 0; t < threads; t++) p->th_.emplace_back([p] { p->worker(); });
        } catch (...) { delete p; return nullptr; }
        cpu_set_t cpus;
        if (node >= 0 && node_cpus(node, &cpus))
            for (auto &x : p->th_) (void)pthread_setaffinity_np(x.native_handle(), sizeof cpus, &cpus);
        return p;
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_.store(true);
        }
        cv_go_.notify_all();
        for (auto &x : th_) if (x.joinable()) x.join();
        release();
    }
    int threads() const { return T_; }
    float *ring(int s) const { return ring_[s]; }
    void start(const double *src, float *dst, size_t n_el)         // asynchronous; n_el <= kChunkEl
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            src_ = src; dst_ = dst; n_el_ = n_el;
            next_.store(0); inexact_.store(0);
            running_.store(T_);
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_go_.notify_all();
    }
    bool wait()                                                     // true: every element of the job was a float32
    {
        // (the calling thread is the critical path: it looks for a short while before it sleeps -- a condition variable's wake-up costs
        //  30 - 50 us, a chunk's copy lasts 280)
        const auto t0 = std::chrono::steady_clock::now();
        while (running_.load(std::memory_order_acquire) != 0) {
            _mm_pause();
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(2 * spin_us_)) {
                std::unique_lock<std::mutex> lk(m_);
                cv_done_.wait(lk, [&] { return running_.load() == 0; });
                break;
            }
        }
        return inexact_.load() == 0;
    }

private:
    Pool() = default;
    void worker()
    {
        uint64_t seen = 0;
        for (;;) {
            const double *src; float *dst; size_t n_el;
            // jobs of one call follow each other every ~0.3 ms: look for the next one for a while before sleeping (the threads stay
            // awake through a call and go to sleep a millisecond after its last chunk)
            const auto t0 = std::chrono::steady_clock::now();
            while (gen_.load(std::memory_order_acquire) == seen && !quit_.load(std::memory_order_relaxed) &&
                   std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(spin_us_)) _mm_pause();
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_go_.wait(lk, [&] { return quit_.load() || gen_.load() != seen; });
                if (quit_.load()) return;
                seen = gen_.load(); src = src_; dst = dst_; n_el = n_el_;
            }
            for (;;) {
                if (inexact_.load(std::memory_order_relaxed)) break;
                const size_t o = next_.fetch_add(1) * kPieceEl;
                if (o >= n_el) break;
                const size_t n = n_el - o < kPieceEl ? n_el - o : kPieceEl;
                const bool exact = avx2_ ? narrow_avx2(src + o, dst + o, n) : narrow_base(src + o, dst + o, n);
                if (!exact) { inexact_.store(1); break; }
            }
            if (running_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(m_);
                cv_done_.notify_one();
            }
        }
    }
    void release()
    {
        for (float *&q : ring_) if (q) { (void)hipHostFree(q); q = nullptr; }
    }
    int T_ = 0;
    bool avx2_ = false;
    int spin_us_ = 0;              // AMX_HOST_SPIN_US (diagnosis): threads that look for the next job before they sleep lost (profiles/r05c_host_transport.txt)
    std::vector<std::thread> th_;
    float *ring_[2] = {nullptr, nullptr};
    std::mutex m_;
    std::condition_variable cv_go_, cv_done_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<int> running_{0};
    std::atomic<bool> quit_{false};
    const double *src_ = nullptr;
    float *dst_ = nullptr;
    size_t n_el_ = 0;
    std::atomic<size_t> next_{0};
    std::atomic<int> inexact_{0};
};

// One host-buffer call's float64 signals, narrowed ahead of the copies: chunk k + 1 (of this batch, or the first one of the NEXT
// batch -- the caller says how long that one is) is being narrowed while the calling thread copies chunk k.
struct Narrower {
    Pool *pool = nullptr;
    const double *base = nullptr;      // the caller's signals
    size_t total_el = 0;
    bool ok = true;                    // false once an element was not a float32: the rest of the call is copied as it is
    bool pending = false;              // a job is in flight (it reads the CALLER's memory: never return without settle())
    size_t p_off = 0, p_n = 0; int p_slot = 0, slot = 0;

    void settle() { if (pending) { (void)pool->wait(); pending = false; } }
    void kick(size_t off, size_t n)
    {
        p_off = off; p_n = n; p_slot = slot; slot ^= 1;
        pool->start(base + off, pool->ring(p_slot), n);
        pending = true;
    }
    // elements [off, off + n) -> dst (device float32).  1: sent, all float32 values -- 0: some element is not (dst holds rubbish,
    // ok = false) -- -1: a copy failed (hip error pending)
    int send(size_t off, size_t n, float *dst, size_t next_batch_el)
    {
        size_t done = 0;
        while (done < n) {
            const size_t len = n - done < Pool::kChunkEl ? n - done : Pool::kChunkEl;
            if (!(pending && p_off == off + done && p_n == len)) { settle(); kick(off + done, len); }
            const int s = p_slot;
            const bool exact = pool->wait();
            pending = false;
            if (!exact) { ok = false; return 0; }
            const size_t n_off = off + done + len;
            const size_t n_len = done + len < n ? (n - done - len < Pool::kChunkEl ? n - done - len : Pool::kChunkEl)
                                                : (next_batch_el < Pool::kChunkEl ? next_batch_el : Pool::kChunkEl);
            if (n_len > 0 && n_off + n_len <= total_el) kick(n_off, n_len);
            if (hipMemcpy(dst + done, pool->ring(s), len * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { settle(); ok = false; return -1; }
            done += len;
        }
        return 1;
    }
};

}  // namespace amx_stage
