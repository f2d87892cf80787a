// Host side: allocator for the expansion's large, short-lived arrays (product code).
// The collector's state vectors, hash tables and op lists grow by doubling and die young; with the
// default allocator every large one is an mmap/munmap pair, and with 16 expansion threads the
// address-space lock and the TLB shoot-downs of those calls cost more than the expansion itself.
// Blocks of 32 KiB and more are therefore rounded to a power of two and recycled through a
// process-wide free list per size; smaller requests go to operator new.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <new>
#include <vector>

#include <sys/mman.h>

namespace tetrex {

class BlockCache {
  public:
    static constexpr size_t kMinBytes = 32u << 10;
    static int size_class(size_t bytes) { return 64 - __builtin_clzll(bytes - 1); }  // ceil(log2)
    static void* take(int cls) {
        Class& c = classes()[cls];
        {
            std::lock_guard<std::mutex> lk(c.m);
            if (!c.free.empty()) {
                void* p = c.free.back();
                c.free.pop_back();
                cached().fetch_sub((size_t)1 << cls, std::memory_order_relaxed);
                return p;
            }
        }
        return fresh((size_t)1 << cls);
    }
    static void give(void* p, int cls) {
        const size_t bytes = (size_t)1 << cls;
        if (cached().load(std::memory_order_relaxed) + bytes > limit()) { std::free(p); return; }
        Class& c = classes()[cls];
        std::lock_guard<std::mutex> lk(c.m);
        c.free.push_back(p);
        cached().fetch_add(bytes, std::memory_order_relaxed);
    }
    // hands cached blocks back to the system until at most `keep` bytes stay cached
    static void trim(size_t keep) {
        for (int cls = 63; cls >= 0 && cached().load() > keep; --cls) {
            Class& c = classes()[cls];
            std::lock_guard<std::mutex> lk(c.m);
            while (!c.free.empty() && cached().load() > keep) {
                std::free(c.free.back());
                c.free.pop_back();
                cached().fetch_sub((size_t)1 << cls);
            }
        }
    }
    static size_t cached_bytes() { return cached().load(); }
    // new memory from the system; blocks of 2 MiB and more ask for transparent huge pages (one page
    // fault per 2 MiB instead of 512: the first touch of a GB of ops is otherwise ~250 000 faults)
    static void* fresh(size_t bytes) {
        void* p = nullptr;
        if (bytes >= kHuge) {
            if (posix_memalign(&p, kHuge, bytes) != 0) throw std::bad_alloc();
            (void)madvise(p, bytes, MADV_HUGEPAGE);
        } else {
            p = std::malloc(bytes);
            if (!p) throw std::bad_alloc();
        }
        return p;
    }
    static constexpr size_t kHuge = (size_t)2 << 20;

  private:
    struct Class { std::mutex m; std::vector<void*> free; };
    struct Registry {  // the cached blocks go back to the system when the process (or the library) goes away
        Class c[64];
        ~Registry() {
            for (Class& k : c) {
                for (void* p : k.free) std::free(p);
                k.free.clear();
            }
        }
    };
    static Class* classes() { static Registry r; return r.c; }
    static std::atomic<size_t>& cached() { static std::atomic<size_t> v{0}; return v; }
    // TETREX_CACHE_MB: upper bound on memory kept for reuse (default 4096)
    static size_t limit() {
        static const size_t v = [] {
            const char* e = std::getenv("TETREX_CACHE_MB");
            const long long mb = e ? std::atoll(e) : 4096;
            return (size_t)(mb < 0 ? 0 : mb) << 20;
        }();
        return v;
    }
};

template <class T>
struct CachedAlloc {
    using value_type = T;
    CachedAlloc() = default;
    template <class U> CachedAlloc(const CachedAlloc<U>&) {}
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes < BlockCache::kMinBytes) return static_cast<T*>(::operator new(bytes));
        return static_cast<T*>(BlockCache::take(BlockCache::size_class(bytes)));
    }
    void deallocate(T* p, size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes < BlockCache::kMinBytes) ::operator delete(p);
        else BlockCache::give(p, BlockCache::size_class(bytes));
    }
    template <class U> bool operator==(const CachedAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const CachedAlloc<U>&) const { return false; }
};

template <class T> using CachedVector = std::vector<T, CachedAlloc<T>>;

}  // namespace tetrex
