// Host side: small open-addressing hash map uint64 -> uint32 (product code).
// Linear probing, power-of-two capacity, no erase; clear() keeps the storage.  Used for the
// collector's per-node state tables and the per-stage k-mer tables, where std::unordered_map's
// node allocations dominated the expansion time.
// Keys below 2^direct_bits (enable_direct) bypass the hash table: they index an array of
// (epoch, value) pairs, so a look-up is one access, clear() is an epoch bump and nothing ever grows —
// the collector's state keys are (k-1) symbols wide, i.e. 16 bits for peptide k = 4.
#pragma once
#include "block_cache.hpp"

#include <cstdint>
#include <utility>
#include <vector>

namespace tetrex {

class FlatMap {
  public:
    FlatMap() = default;
    size_t size() const { return size_ + direct_size_; }
    size_t capacity() const { return cap_ + direct_.size(); }
    // keys < 2^bits (bits <= 20) may move to a directly indexed array once the map is large (reserve)
    void want_direct(unsigned bits) { wanted_bits_ = bits <= 20 ? bits : 0; }
    void clear() {
        if (direct_bits_ && direct_size_) {
            direct_size_ = 0;
            if (++epoch_ == 0) { direct_.assign(direct_.size(), Direct{0, 0}); epoch_ = 1; }
        }
        if (size_ == 0) return;
        if (cap_ > 1024 && size_ * 8 < cap_) {  // shrink tables that were briefly huge
            slots_.clear(); slots_.shrink_to_fit();
            cap_ = 0;
        } else {
            for (Slot& s : slots_) s.val = kEmpty;
        }
        size_ = 0;
    }
    // returns (value slot, inserted); a new entry gets `value`
    std::pair<uint32_t*, bool> emplace(uint64_t key, uint32_t value) {
        if (direct_bits_ && (key >> direct_bits_) == 0) {
            Direct& d = direct_[key];
            if (d.epoch == epoch_) return {&d.val, false};
            d.epoch = epoch_;
            d.val = value;
            ++direct_size_;
            return {&d.val, true};
        }
        if ((size_ + 1) * 4 > cap_ * 3) grow();
        size_t i = mix(key) & (cap_ - 1);
        for (;;) {
            Slot& s = slots_[i];
            if (s.val == kEmpty) {
                s.key = key;
                s.val = value;
                ++size_;
                return {&s.val, true};
            }
            if (s.key == key) return {&s.val, false};
            i = (i + 1) & (cap_ - 1);
        }
    }
    // pulls the cache line a later emplace(key, ...) will look at first
    void prefetch(uint64_t key) const {
        if (direct_bits_ && (key >> direct_bits_) == 0) __builtin_prefetch(&direct_[key]);
        else if (cap_) __builtin_prefetch(&slots_[mix(key) & (cap_ - 1)]);
    }
    // room for `n` entries without growing
    void reserve(size_t n) {
        if (wanted_bits_ && !direct_bits_ && n >= kDirectFrom) go_direct();
        if (direct_bits_) return;  // the large key population is in the array
        size_t want = 16;
        while (want * 3 < n * 4) want <<= 1;
        if (want > cap_) grow(want);
    }

  private:
    static constexpr uint32_t kEmpty = 0xFFFFFFFFu;  // values must never be 0xFFFFFFFF
    struct Slot { uint64_t key; uint32_t val; uint32_t pad; };  // key and value share a cache line: one miss per probe
    CachedVector<Slot> slots_;
    size_t cap_ = 0, size_ = 0;
    static constexpr size_t kDirectFrom = 2048;  // entries from which the 2^bits-entry array is worth its memory
    struct Direct { uint32_t epoch, val; };
    CachedVector<Direct> direct_;
    unsigned direct_bits_ = 0, wanted_bits_ = 0;
    uint32_t epoch_ = 1;
    size_t direct_size_ = 0;

    static uint64_t mix(uint64_t x) {
        x ^= x >> 33;
        x *= 0xff51afd7ed558ccdULL;
        x ^= x >> 33;
        return x;
    }
    // allocate the array and move the small keys over; the (few) large ones are re-hashed
    void go_direct() {
        direct_bits_ = wanted_bits_;
        direct_.assign((size_t)1 << direct_bits_, Direct{0, 0});
        epoch_ = 1;
        direct_size_ = 0;
        CachedVector<Slot> old;
        old.swap(slots_);
        cap_ = 0;
        size_ = 0;
        for (const Slot& s : old)
            if (s.val != kEmpty) emplace(s.key, s.val);
    }
    void grow(size_t to = 0) {
        const size_t ncap = to ? to : (cap_ ? cap_ * 2 : 16);
        CachedVector<Slot> old;
        old.swap(slots_);
        slots_.assign(ncap, Slot{0, kEmpty, 0});
        cap_ = ncap;
        size_ = 0;
        for (const Slot& s : old)
            if (s.val != kEmpty) emplace(s.key, s.val);
    }
};

}  // namespace tetrex
