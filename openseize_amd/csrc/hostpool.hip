// hostpool.hip -- host side of host-fed streams: a pitched 2-D copy over a few persistent
// threads.  The reference's sources are host ndarrays (core/producer.py:289-295: a chunk is a
// column range of a C-ordered array, i.e. rows `pitch` bytes apart); before it can leave on
// an H2D stream it has to be packed into a pinned staging buffer, and at the reference's own
// chunk size (30 000 samples, cfg-1) that packing IS the per-chunk cost of the host-fed path:
// Python's thread pool spends 0.3 ms per 3.8 MB chunk mostly on waking its workers.  Here
// the workers sleep on a condition variable between jobs, a job is a handful of memcpy calls per
// worker (the caller takes its share), and the call returns when the rows are in place.
//
// fork(): the workers exist in the process that created them only.  The pool is never destroyed
// (no join at exit: a forked child that runs static destructors would wait for threads it does
// not have), and a child made by fork() copies single-threaded: pthread_atfork marks the pool it
// inherited as without workers, and a pool without workers never touches its locks (which the
// parent may have held when it forked).
#include <pthread.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

namespace osz {

class CopyPool {
  public:
    static CopyPool &get() {
        static CopyPool *p = [] {
            CopyPool *q = new CopyPool();           // leaked on purpose, see above
            pthread_atfork(nullptr, nullptr, [] { child_ = true; });
            return q;
        }();
        return *p;
    }
    int workers() const { return child_ ? 0 : (int)threads_.size(); }

    // rows of row_bytes bytes from src (pitch sp) to dst (pitch dp), split over the workers
    // and the caller
    void copy2d(char *dst, int64_t dp, const char *src, int64_t sp, int64_t rows, int64_t row_bytes) {
        const int64_t total = rows * row_bytes;
        int parts = (int)(total / (512 << 10));                 // at least 512 KB per part
        if (parts > workers() + 1) parts = workers() + 1;
        if (parts <= 1) {      // (also every copy of a forked child: it has no workers and touches no lock)
            run_part(dst, dp, src, sp, rows, row_bytes, 0, 1);
            return;
        }
        std::lock_guard<std::mutex> one_job(submit_);           // one job at a time
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = Job{dst, dp, src, sp, rows, row_bytes, parts};
            ++epoch_;
            done_.store(0, std::memory_order_relaxed);
            ticket_.store((epoch_ << 32) | 1u, std::memory_order_release);   // part 0 is the caller's
        }
        const uint64_t ep = epoch_;
        cv_.notify_all();
        run_part(dst, dp, src, sp, rows, row_bytes, 0, parts);
        // help with what is left, then wait for the stragglers
        for (int p; claim(ep, parts, p);) {
            run_part(dst, dp, src, sp, rows, row_bytes, p, parts);
            done_.fetch_add(1, std::memory_order_release);
        }
        while (done_.load(std::memory_order_acquire) < parts - 1) std::this_thread::yield();
    }

  private:
    struct Job {
        char *dst;
        int64_t dp;
        const char *src;
        int64_t sp, rows, row_bytes;
        int parts;
    };

    CopyPool() {
        unsigned hw = std::thread::hardware_concurrency();
        int n = hw >= 16 ? 7 : hw >= 8 ? 3 : hw >= 4 ? 1 : 0;
        for (int i = 0; i < n; ++i) {
            threads_.emplace_back([this] { loop(); });
            threads_.back().detach();
        }
    }

    // A part is claimed together with the job's number: a worker that comes back late from the
    // previous job finds the ticket re-issued and claims nothing of a job it knows nothing of.
    bool claim(uint64_t ep, int parts, int &p) {
        uint64_t v = ticket_.load(std::memory_order_acquire);
        for (;;) {
            if ((v >> 32) != (ep & 0xffffffffu)) return false;
            const int part = (int)(uint32_t)v;
            if (part >= parts) return false;
            if (ticket_.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel)) {
                p = part;
                return true;
            }
        }
    }

    // part p of `parts`: a contiguous range of the (row, byte) space, cut on 64-byte lines
    static void run_part(char *dst, int64_t dp, const char *src, int64_t sp, int64_t rows, int64_t row_bytes,
                         int p, int parts) {
        if (rows >= parts) {
            const int64_t r0 = rows * p / parts, r1 = rows * (p + 1) / parts;
            if (dp == row_bytes && sp == row_bytes) {
                std::memcpy(dst + r0 * dp, src + r0 * sp, (size_t)((r1 - r0) * row_bytes));
                return;
            }
            for (int64_t r = r0; r < r1; ++r) std::memcpy(dst + r * dp, src + r * sp, (size_t)row_bytes);
        } else {
            const int64_t b0 = (row_bytes * p / parts) & ~(int64_t)63, b1 = p + 1 == parts ? row_bytes : (row_bytes * (p + 1) / parts) & ~(int64_t)63;
            for (int64_t r = 0; r < rows; ++r) std::memcpy(dst + r * dp + b0, src + r * sp + b0, (size_t)(b1 - b0));
        }
    }

    void loop() {
        uint64_t seen = 0;
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return epoch_ != seen; });
                seen = epoch_;
                j = job_;
            }
            for (int p; claim(seen, j.parts, p);) {
                run_part(j.dst, j.dp, j.src, j.sp, j.rows, j.row_bytes, p, j.parts);
                done_.fetch_add(1, std::memory_order_release);
            }
        }
    }

    static inline bool child_ = false;    // set in a fork()ed child: no workers there
    std::vector<std::thread> threads_;
    std::mutex m_, submit_;
    std::condition_variable cv_;
    Job job_{};
    uint64_t epoch_ = 0;
    std::atomic<uint64_t> ticket_{0};   // (job number << 32) | next part
    std::atomic<int> done_{0};
};

}  // namespace osz

extern "C" int osz_host_copy2d(void *dst, int64_t dst_pitch, const void *src, int64_t src_pitch, int64_t rows,
                               int64_t row_bytes) {
    OSZ_REQUIRE(rows >= 0 && row_bytes >= 0, "osz_host_copy2d: rows=%lld row_bytes=%lld", (long long)rows,
                (long long)row_bytes);
    if (rows == 0 || row_bytes == 0) return OSZ_OK;
    OSZ_REQUIRE(dst && src && dst_pitch >= row_bytes && src_pitch >= row_bytes,
                "osz_host_copy2d: pitches %lld / %lld for rows of %lld bytes", (long long)dst_pitch,
                (long long)src_pitch, (long long)row_bytes);
    osz::CopyPool::get().copy2d(static_cast<char *>(dst), dst_pitch, static_cast<const char *>(src), src_pitch, rows,
                                row_bytes);
    return OSZ_OK;
}
