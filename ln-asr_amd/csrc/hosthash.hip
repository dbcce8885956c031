// Host-side fingerprint of model parameters (no device code).
//
// The reference's methods re-read self.A / self.mu / self.si on every call (hmm.py:82-99, gmmhmm.py:58-67), so the
// mirror classes must notice in-place edits between calls: they fingerprint the parameter arrays and upload only when
// the fingerprint changed.  For the metric model that is 25 MB of `si` per decode() call; one host thread hashes it in
// ~1.5 ms, which would be half of the call.  lnasr_hash128 cuts the buffer into segments hashed by a small persistent
// pool of host threads (4 x 64-bit multiply-rotate lanes per segment, the segment digests folded in order), so the
// check costs ~0.2 ms and does not depend on an optional Python module.
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "common.h"

namespace lnasr {
namespace {

constexpr uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull,
                   P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;

inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t rd64(const unsigned char* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}
inline uint64_t lane(uint64_t acc, uint64_t w) { return rotl(acc + w * P2, 31) * P1; }
inline uint64_t avalanche(uint64_t h) {
    h ^= h >> 33;
    h *= P2;
    h ^= h >> 29;
    h *= P3;
    h ^= h >> 32;
    return h;
}

// Two 64-bit digests of one segment.  Each 64-bit lane update is a bijection of the word it absorbs and of the lane
// state, so a change confined to one lane always changes that lane's final state and both digests; changes spread
// over several lanes would have to cancel in two differently mixed folds of the four lanes at once.
void hash_segment(const unsigned char* p, size_t n, uint64_t seed, uint64_t out[2]) {
    uint64_t a0 = seed + P1 + P2, a1 = seed + P2, a2 = seed, a3 = seed - P1;
    size_t left = n;                     // counted, not compared as pointers: p may be null when n == 0
    for (; left >= 32; left -= 32, p += 32) {
        a0 = lane(a0, rd64(p));
        a1 = lane(a1, rd64(p + 8));
        a2 = lane(a2, rd64(p + 16));
        a3 = lane(a3, rd64(p + 24));
    }
    uint64_t h = rotl(a0, 1) + rotl(a1, 7) + rotl(a2, 12) + rotl(a3, 18);
    uint64_t g = (rotl(a0, 43) * P3) ^ (rotl(a1, 31) * P4) ^ (rotl(a2, 19) * P5) ^ (a3 * P2);
    h += (uint64_t)n;
    g ^= (uint64_t)n * P5;
    for (; left >= 8; left -= 8, p += 8) {
        const uint64_t w = rd64(p);
        h = rotl(h ^ lane(0, w), 27) * P1 + P4;
        g = (rotl(g, 23) ^ w) * P2 + P3;
    }
    for (; left > 0; --left, ++p) {
        h = rotl(h ^ (*p * P5), 11) * P1;
        g = (rotl(g, 17) ^ *p) * P3;
    }
    out[0] = avalanche(h);
    out[1] = avalanche(g ^ (g >> 31));
}

// A few persistent host threads (created on first use, parked on a condition variable in between).
class Pool {
  public:
    static Pool& get() {
        static Pool* p = new Pool();         // never destroyed: worker threads may outlive static destructors otherwise
        return *p;
    }
    int size() const { return (int)threads_.size() + 1; }
    // runs fn(i) for i in [0, n) on the pool plus the calling thread; returns when all are done
    void run(int n, const std::function<void(int)>& fn) {
        std::lock_guard<std::mutex> serial(run_mu_);          // one parallel region at a time
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            n_ = n;
            next_ = 0;
            pending_ = n;
            ++gen_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
        n_ = 0;
    }

  private:
    Pool() {
        unsigned hw = std::thread::hardware_concurrency();
        int n = hw == 0 ? 4 : (int)hw;
        if (const char* e = getenv("LNASR_HASH_THREADS")) n = atoi(e);
        n = n < 1 ? 1 : (n > 8 ? 8 : n);
        for (int t = 1; t < n; ++t) threads_.emplace_back([this] { loop(); });
        for (auto& t : threads_) t.detach();
    }
    void work() {
        // every field is read and written under mu_: a worker that wakes late simply joins the region that is current
        std::unique_lock<std::mutex> lk(mu_);
        while (next_ < n_) {
            const int i = next_++;
            const std::function<void(int)>* fn = fn_;
            lk.unlock();
            (*fn)(i);
            lk.lock();
            if (--pending_ == 0) done_cv_.notify_all();
        }
    }
    void loop() {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
            }
            work();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int)>* fn_ = nullptr;
    int n_ = 0, pending_ = 0;
    int next_ = 0;
    unsigned long gen_ = 0;
};

}  // namespace
}  // namespace lnasr

using namespace lnasr;

namespace {
// one background fingerprint of several buffers (lnasr_hash128_begin / _end)
struct HashJob {
    std::vector<const void*> ptr;
    std::vector<size_t> bytes;
    std::vector<uint64_t> out;           // two words per buffer
    uint64_t seed = 0;
    int rc = 0;
    std::thread th;
};
}  // namespace

extern "C" {

int lnasr_hash128(const void* data, size_t bytes, uint64_t seed, uint64_t* out2) {
    LN_REQUIRE(out2, "null out");
    LN_REQUIRE(data || bytes == 0, "null data");
    const unsigned char* p = static_cast<const unsigned char*>(data);
    constexpr size_t kSeg = 1u << 20;                       // below 2 segments: the calling thread alone
    if (bytes < 2 * kSeg) {
        hash_segment(p, bytes, seed, out2);
        return 0;
    }
    Pool& pool = Pool::get();
    // the segmentation depends on the size only (never on the thread count): a fingerprint is comparable across machines
    int nseg = (int)(bytes / kSeg);                         // ~1 MB pieces, handed out dynamically
    nseg = nseg < 2 ? 2 : (nseg > 64 ? 64 : nseg);
    const size_t per = ((bytes / nseg) + 63) & ~(size_t)63;
    std::vector<uint64_t> dig((size_t)nseg * 2);
    pool.run(nseg, [&](int i) {
        const size_t lo = (size_t)i * per, hi = i + 1 == nseg ? bytes : (lo + per < bytes ? lo + per : bytes);
        hash_segment(p + (lo < bytes ? lo : bytes), hi > lo ? hi - lo : 0, seed + (uint64_t)i * P5, &dig[(size_t)i * 2]);
    });
    hash_segment(reinterpret_cast<const unsigned char*>(dig.data()), dig.size() * 8, seed ^ (uint64_t)bytes, out2);
    return 0;
}

// The same fingerprint of n buffers (buffer k hashed with seed + k) computed in the BACKGROUND: _begin returns at once,
// _end waits and writes two words per buffer.  The mirror classes start it on the model parameters, enqueue the call's
// kernels on the model the device already holds while it runs, and keep the result only if the fingerprint confirms
// that nothing changed (lnasr/gmmhmm.py: the 0.2 ms check of a 25 MB model then costs the call nothing).  The buffers
// must stay alive and unchanged until _end; every job begun must be ended (it owns a thread).
int lnasr_hash128_begin(const void* const* ptrs, const size_t* bytes, int n, uint64_t seed, void** job_out) {
    LN_REQUIRE(job_out, "null job_out");
    *job_out = nullptr;
    LN_REQUIRE(n >= 0 && (n == 0 || (ptrs && bytes)), "bad buffer list");
    for (int k = 0; k < n; ++k) LN_REQUIRE(ptrs[k] || bytes[k] == 0, "null buffer %d", k);
    HashJob* j = new HashJob();
    j->ptr.assign(ptrs, ptrs + n);
    j->bytes.assign(bytes, bytes + n);
    j->out.assign((size_t)2 * n, 0);
    j->seed = seed;
    try {
        j->th = std::thread([j] {
            for (size_t k = 0; k < j->ptr.size() && j->rc == 0; ++k)
                j->rc = lnasr_hash128(j->ptr[k], j->bytes[k], j->seed + k, &j->out[2 * k]);
        });
    } catch (...) {
        delete j;
        return fail(LNASR_E_INVALID, "could not start the fingerprint thread");
    }
    *job_out = j;
    return 0;
}

int lnasr_hash128_end(void* job, uint64_t* out) {
    LN_REQUIRE(job, "null job");
    HashJob* j = static_cast<HashJob*>(job);
    if (j->th.joinable()) j->th.join();
    const int rc = j->rc;
    if (rc == 0 && out)
        for (size_t k = 0; k < j->out.size(); ++k) out[k] = j->out[k];
    delete j;
    return rc;
}

}  // extern "C"
