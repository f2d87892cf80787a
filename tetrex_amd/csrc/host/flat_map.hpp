// Host side: small open-addressing hash map uint64 -> uint32 (product code).
// Linear probing, power-of-two capacity, no erase; clear() keeps the storage.  Used for the
// collector's per-node state tables and the per-stage k-mer tables, where std::unordered_map's
// node allocations dominated the expansion time.
#pragma once
#include "block_cache.hpp"

#include <cstdint>
#include <utility>
#include <vector>

namespace tetrex {

class FlatMap {
  public:
    FlatMap() = default;
    size_t size() const { return size_; }
    size_t capacity() const { return cap_; }
    void clear() {
        if (size_ == 0) return;
        if (cap_ > 1024 && size_ * 8 < cap_) {  // shrink tables that were briefly huge
            keys_.clear(); keys_.shrink_to_fit();
            vals_.clear(); vals_.shrink_to_fit();
            cap_ = 0;
        } else {
            std::fill(vals_.begin(), vals_.end(), kEmpty);
        }
        size_ = 0;
    }
    // returns (value slot, inserted); a new entry gets `value`
    std::pair<uint32_t*, bool> emplace(uint64_t key, uint32_t value) {
        if ((size_ + 1) * 4 > cap_ * 3) grow();
        size_t i = mix(key) & (cap_ - 1);
        for (;;) {
            if (vals_[i] == kEmpty) {
                keys_[i] = key;
                vals_[i] = value;
                ++size_;
                return {&vals_[i], true};
            }
            if (keys_[i] == key) return {&vals_[i], false};
            i = (i + 1) & (cap_ - 1);
        }
    }

  private:
    static constexpr uint32_t kEmpty = 0xFFFFFFFFu;  // values must never be 0xFFFFFFFF
    CachedVector<uint64_t> keys_;
    CachedVector<uint32_t> vals_;
    size_t cap_ = 0, size_ = 0;

    static uint64_t mix(uint64_t x) {
        x ^= x >> 33;
        x *= 0xff51afd7ed558ccdULL;
        x ^= x >> 33;
        return x;
    }
    void grow() {
        const size_t ncap = cap_ ? cap_ * 2 : 16;
        CachedVector<uint64_t> ok;
        CachedVector<uint32_t> ov;
        ok.swap(keys_);
        ov.swap(vals_);
        keys_.assign(ncap, 0);
        vals_.assign(ncap, kEmpty);
        const size_t ocap = cap_;
        cap_ = ncap;
        size_ = 0;
        for (size_t i = 0; i < ocap; ++i)
            if (ov[i] != kEmpty) emplace(ok[i], ov[i]);
    }
};

}  // namespace tetrex
