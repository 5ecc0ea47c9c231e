// amx_stage.hpp -- float64 signals in pageable host memory travel to the device as float32 when that loses nothing.
//
// The reference's boundary is host numpy in, host numpy out (core.py:465-466), and evaluation.y is the float64 CAST of a
// float32 image (core.py:136 loads niiDWI_img as float32, core.py:209-223 normalises it in place, models.pyx:902 reads the
// float64 copy): every value of y is exactly a float32.  A host-buffer fit of 1 M NODDI voxels is 14.9 ms of PCIe
// (792 MB at the link's 56.4 GB/s, measured per batch inside the call: AMX_HOST_TRACE=1, profiles/r05c_host_trace.txt) next
// to 6.5 ms of kernels -- the copy IS the call.  So the copy is halved where that is exact: host threads narrow the caller's
// buffer chunk by chunk into a ring of four pinned buffers, CHECKING every element ((double)(float)v == v; 144 - 260 GB/s of
// float64 on the box's cores, tools/probes/h2d_probe.hip), the calling thread sends chunk k with the blocking copy it always
// used while the chunks behind it are being narrowed, and the device widens the batch again (k_widen, amx_api.hip).  One element that is not a
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
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <cstdio>
#include <pthread.h>
#include <sched.h>
#include <unistd.h>

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

// physical cores in a CPU set (the first hardware thread of each core that lies in the set)
static int count_phys(const cpu_set_t &cpus)
{
    int n = 0;
    for (int k = 0; k < CPU_SETSIZE; k++) {
        if (!CPU_ISSET(k, &cpus)) continue;
        char path[128];
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", k);
        int first = k;
        if (FILE *f = fopen(path, "r")) { if (fscanf(f, "%d", &first) != 1) first = k; fclose(f); }
        if (first == k || !CPU_ISSET(first, &cpus)) n++;
    }
    return n;
}
// which of the devices that hang on `node` is `device` (by device number), and how many there are: the siblings of its pool
static void device_siblings(int device, int node, int *idx, int *count)
{
    *idx = 0; *count = 1;
    int nd = 0;
    if (node < 0 || hipGetDeviceCount(&nd) != hipSuccess || nd <= 1) { (void)hipGetLastError(); return; }
    int n = 0, i = 0;
    for (int d = 0; d < nd; d++) {
        if (device_node(d) != node) continue;
        if (d == device) i = n;
        n++;
    }
    if (n > 1) { *idx = i; *count = n; return; }
    // one visible device per process (HIP_VISIBLE_DEVICES set by the launcher): the launcher's local rank / size say who else is there
    const char *lr = getenv("LOCAL_RANK"), *lw = getenv("LOCAL_WORLD_SIZE");
    if (lr && lw && atoi(lw) > 1) {
        int nodes = 0;
        cpu_set_t cs;
        for (int k = 0; k < 16; k++) if (node_cpus(k, &cs)) nodes++;
        const int per = (atoi(lw) + (nodes > 0 ? nodes : 1) - 1) / (nodes > 0 ? nodes : 1);
        *count = per > 1 ? per : 1; *idx = atoi(lr) % *count;
    }
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

// T host threads that narrow ONE CALL's signals, chunk by chunk, into a ring of pinned buffers, ahead of the calling thread's copies.
// The job is the whole call (its chunks are known up front: the batches of fit_host are), so the threads wake once per call and run
// until the ring is full (kRing chunks = 1.1 ms of link ahead of the copies) -- a chunk-at-a-time hand-over (one wake-up per chunk,
// one chunk of slack) left the calling thread waiting for the narrowing whenever the threads sat on the far socket or woke late:
// 12.5 ms per 1 M voxels in one process, 14 - 16 ms in another (profiles/r05c_host_transport.txt, section 8).
struct Chunk { size_t off, n; };       // elements of the caller's buffer

class Pool {
public:
    static constexpr size_t kChunkEl = 4u << 20;           // elements per chunk: 32 MB read, 16 MB sent (0.28 ms of link)
    static constexpr size_t kPieceEl = 32u << 10;          // elements a thread takes at a time
    static constexpr int kRing = 4;
    // node >= 0: the threads stay on that NUMA node's CPUs.  sib_n > 1 (round 6): this pool is one of sib_n on the node -- one per device that
    // hangs on it, whether the devices are driven by one process (models.py: AMX_DEVICES) or by one process each (torchrun) -- and takes the
    // sib_i-th share of the node's physical cores: stripes of its own, disjoint from its siblings' (twelve threads per
    // pool striped over ALL the node's cores overlap pairwise as soon as two pools share a socket).  A share too small for `threads` stripes
    // of two cores gets fewer threads (never below four: the narrowing of 113 GB/s of float64 needs them).
    static int sibling_threads(int threads, int node_phys_cores, int sib_n)
    {
        if (sib_n <= 1 || node_phys_cores <= 0) return threads;
        const int share = node_phys_cores / sib_n;
        int t = threads;
        if (share < 2 * t) t = share / 2;
        return t < 4 ? (share >= 4 ? 4 : (share > 0 ? share : 1)) : t;
    }
    static Pool *create(int threads, int node = -1, int sib_i = 0, int sib_n = 1)
    {
        if (node >= 0 && sib_n > 1) {
            cpu_set_t c0;
            if (node_cpus(node, &c0)) threads = sibling_threads(threads, count_phys(c0), sib_n);
        }
        Pool *p = new Pool();
        p->T_ = threads;
        p->avx2_ = __builtin_cpu_supports("avx2");
        // the pinned ring is allocated (and its pages placed) by a thread that sits on the DEVICE's node for the length of the allocation: the
        // narrowing threads write it from there and the copies leave from there -- whoever makes the pool (the calling thread of a first fit, or,
        // round 6, a helper thread beside the dictionary upload) may be running on the other socket
        cpu_set_t before, on_node;
        const bool moved = node >= 0 && node_cpus(node, &on_node) && pthread_getaffinity_np(pthread_self(), sizeof before, &before) == 0 &&
                           pthread_setaffinity_np(pthread_self(), sizeof on_node, &on_node) == 0;
        bool ring_ok = true;
        for (float *&q : p->ring_)
            if (hipHostMalloc((void **)&q, kChunkEl * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); q = nullptr; ring_ok = false; break; }
        if (ring_ok) for (float *q : p->ring_) for (size_t e = 0; e < kChunkEl; e += 1024) q[e] = 0.0f;       // (first touch, should the runtime leave it to us)
        if (moved) (void)pthread_setaffinity_np(pthread_self(), sizeof before, &before);
        if (!ring_ok) { p->release(); delete p; return nullptr; }
        try {
            for (int t = 0; t < threads; t++) p->th_.emplace_back([p] { p->worker(); });
        } catch (...) { delete p; return nullptr; }
        // Every thread gets its own STRIPE of the node's physical cores (P / T of them; AMX_HOST_PIN_CORES=0: all threads share the node's
        // CPUs as one set, 1: one core per thread -- diagnosis).  Threads that wake together start on the waker's cache domain and are
        // spread by the load balancer over MILLISECONDS: with the node as one set the first batches of a call were narrowed at a third of
        // the rate of the later ones.  One core per thread cures that and has a failure of its own: a thread nailed to the core the CALLING
        // thread happens to run on shares it with that thread's copies -- 28 ms per call instead of 12.6, one process in five.  Within a
        // stripe the scheduler can step aside (section 10 of profiles/r05c_host_transport.txt).  The stripes rotate with the process id.
        cpu_set_t cpus;
        if (node >= 0 && node_cpus(node, &cpus)) {
            std::vector<int> phys;
            for (int k = 0; k < CPU_SETSIZE; k++) {
                if (!CPU_ISSET(k, &cpus)) continue;
                char path[128];
                snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", k);
                int first = k;
                if (FILE *f = fopen(path, "r")) { if (fscanf(f, "%d", &first) != 1) first = k; fclose(f); }
                if (first == k || !CPU_ISSET(first, &cpus)) phys.push_back(k);
            }
            const char *pc = getenv("AMX_HOST_PIN_CORES");
            if (sib_n > 1 && (int)phys.size() >= sib_n) {
                // this pool's share of the node: cores [sib_i * share, (sib_i + 1) * share) in the node's order
                const int share = (int)phys.size() / sib_n;
                phys = std::vector<int>(phys.begin() + (size_t)(sib_i % sib_n) * share, phys.begin() + (size_t)(sib_i % sib_n + 1) * share);
            }
            const int P = (int)phys.size();
            p->share_first_ = P > 0 ? phys.front() : -1; p->share_cores_ = P;
            if ((!pc || pc[0] != '0') && P >= 2 * threads) {
                // (a pool alone on its node rotates its stripes with the process id -- processes that do not know of each other; a
                //  sibling's share is its own: no rotation)
                const int stride = P / threads, start = sib_n > 1 ? 0 : (int)(((unsigned)getpid() * 7u) % (unsigned)P);
                const int width = (pc && pc[0] == '1') ? 1 : stride;
                for (int t = 0; t < threads; t++) {
                    cpu_set_t mine;
                    CPU_ZERO(&mine);
                    for (int j = 0; j < width; j++) CPU_SET(phys[(start + t * stride + j) % P], &mine);
                    (void)pthread_setaffinity_np(p->th_[t].native_handle(), sizeof mine, &mine);
                }
            } else if (sib_n > 1 && P > 0) {
                cpu_set_t mine;
                CPU_ZERO(&mine);
                for (int c : phys) CPU_SET(c, &mine);
                for (auto &x : p->th_) (void)pthread_setaffinity_np(x.native_handle(), sizeof mine, &mine);
            } else {
                for (auto &x : p->th_) (void)pthread_setaffinity_np(x.native_handle(), sizeof cpus, &cpus);
            }
        }
        return p;
    }
    ~Pool()
    {
        end();
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_.store(true);
        }
        cv_go_.notify_all();
        for (auto &x : th_) if (x.joinable()) x.join();
        release();
    }
    int threads() const { return T_; }
    int share_first() const { return share_first_; }       // first CPU and number of physical cores of this pool's share of its node (diagnosis, tests)
    int share_cores() const { return share_cores_; }
    float *slot(int c) const { return ring_[c % kRing]; }

    // start narrowing base[chunks[0]], base[chunks[1]], ... (every n <= kChunkEl); the threads read the CALLER's memory until end()
    void begin(const double *base, const std::vector<Chunk> &chunks)
    {
        end();
        {
            std::lock_guard<std::mutex> lk(m_);
            base_ = base; chunks_ = chunks;
            const int nc = (int)chunks_.size();
            first_.assign(nc + 1, 0);
            for (int c = 0; c < nc; c++) first_[c + 1] = first_[c] + (chunks_[c].n + kPieceEl - 1) / kPieceEl;
            done_.reset(new std::atomic<int>[nc > 0 ? nc : 1]);
            for (int c = 0; c < nc; c++) done_[c].store(0);
            next_.store(0); copied_.store(0); inexact_.store(false); abort_.store(false);
            running_.store(T_); active_ = true;
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_go_.notify_all();
    }
    // blocks until chunk c lies in slot(c): true -- false: an element of the call was not a float32 (end() the job, copy the rest plainly)
    bool ready(int c)
    {
        const int want = (int)(first_[c + 1] - first_[c]);
        for (int spin = 0; done_[c].load(std::memory_order_acquire) < want; spin++) {
            if (inexact_.load(std::memory_order_relaxed)) return false;
            nap(spin);
        }
        return true;
    }
    void consumed(int c) { copied_.store(c + 1, std::memory_order_release); }       // slot(c) may be overwritten
    // stop whatever is left of the job and wait until no thread reads the caller's memory any more
    void end()
    {
        if (!active_) return;
        abort_.store(true);
        for (int spin = 0; running_.load(std::memory_order_acquire) != 0; spin++) nap(spin);
        active_ = false;
    }

private:
    Pool() = default;
    static void nap(int spin)
    {
        if (spin < 64) _mm_pause();
        else std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    void worker()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_go_.wait(lk, [&] { return quit_.load() || gen_.load() != seen; });
                if (quit_.load()) return;
                seen = gen_.load();
            }
            const size_t total = first_.back();
            const int nc = (int)chunks_.size();
            int c = 0;
            for (;;) {
                if (inexact_.load(std::memory_order_relaxed) || abort_.load(std::memory_order_relaxed)) break;
                const size_t p = next_.fetch_add(1);
                if (p >= total) break;
                while (c + 1 < nc && p >= first_[c + 1]) c++;
                bool stop = false;
                for (int spin = 0; c >= copied_.load(std::memory_order_acquire) + kRing; spin++) {      // the ring is full: the copies decide the pace
                    if (inexact_.load(std::memory_order_relaxed) || abort_.load(std::memory_order_relaxed)) { stop = true; break; }
                    nap(spin);
                }
                if (stop) break;
                const size_t o = (p - first_[c]) * kPieceEl;
                const size_t n = chunks_[c].n - o < kPieceEl ? chunks_[c].n - o : kPieceEl;
                const double *src = base_ + chunks_[c].off + o;
                float *dst = ring_[c % kRing] + o;
                const bool exact = avx2_ ? narrow_avx2(src, dst, n) : narrow_base(src, dst, n);
                if (!exact) { inexact_.store(true); break; }
                done_[c].fetch_add(1, std::memory_order_release);
            }
            running_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    void release()
    {
        for (float *&q : ring_) if (q) { (void)hipHostFree(q); q = nullptr; }
    }
    int T_ = 0;
    int share_first_ = -1, share_cores_ = 0;
    bool avx2_ = false;
    bool active_ = false;          // (calling thread only)
    std::vector<std::thread> th_;
    float *ring_[kRing] = {nullptr, nullptr, nullptr, nullptr};
    std::mutex m_;
    std::condition_variable cv_go_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<int> running_{0};
    std::atomic<bool> quit_{false};
    const double *base_ = nullptr;
    std::vector<Chunk> chunks_;
    std::vector<size_t> first_;                     // pieces before chunk c
    std::unique_ptr<std::atomic<int>[]> done_;      // narrowed pieces of chunk c
    std::atomic<size_t> next_{0};
    std::atomic<int> copied_{0};                    // chunks whose slots the calling thread has sent
    std::atomic<bool> inexact_{false}, abort_{false};
};

}  // namespace amx_stage
