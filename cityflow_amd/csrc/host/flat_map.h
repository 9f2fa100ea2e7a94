// Open-addressing int32 -> int32 hash map (linear probing, tombstones, power-of-two capacity).
// The spawner inserts one entry per spawned vehicle and never shrinks, so at city scale the table is tens of MiB and
// every access is a cache miss: key and value share one 8-byte entry (one miss per access, state folded into the value)
// and prefetch() lets the caller overlap the misses of a whole step.
#pragma once

#include <cstdint>
#include <vector>

namespace cfa {

class FlatMapI32 {
public:
    FlatMapI32() { rehash(1024); }

    // values must be > kTomb (the spawner stores vehicle ids and -1)
    int32_t *find(int32_t key) {
        size_t i = slotOf(key);
        for (;;) {
            Entry &e = entries_[i];
            if (e.val == kEmpty) return nullptr;
            if (e.val != kTomb && e.key == key) return &e.val;
            i = (i + 1) & mask_;
        }
    }
    void set(int32_t key, int32_t value) {
        if ((used_ + 1) * 10 > (mask_ + 1) * 6) rehash((mask_ + 1) * 2);
        size_t i = slotOf(key), firstTomb = (size_t) -1;
        for (;;) {
            Entry &e = entries_[i];
            if (e.val == kEmpty) break;
            if (e.val == kTomb) {
                if (firstTomb == (size_t) -1) firstTomb = i;
            } else if (e.key == key) {
                e.val = value;
                return;
            }
            i = (i + 1) & mask_;
        }
        if (firstTomb != (size_t) -1) i = firstTomb;
        else ++used_;
        entries_[i].key = key;
        entries_[i].val = value;
        ++size_;
    }
    void erase(int32_t key) {
        size_t i = slotOf(key);
        for (;;) {
            Entry &e = entries_[i];
            if (e.val == kEmpty) return;
            if (e.val != kTomb && e.key == key) {
                e.val = kTomb;
                --size_;
                return;
            }
            i = (i + 1) & mask_;
        }
    }
    // make room for n entries without a rehash (a rehash of a multi-million entry table stalls the caller for ~100 ms)
    void reserve(size_t n) {
        size_t cap = mask_ + 1;
        while (n * 10 > cap * 6) cap *= 2;
        if (cap != mask_ + 1) rehash(cap);
    }
    void prefetch(int32_t key) const { __builtin_prefetch(&entries_[slotOf(key)], 1, 1); }
    void clear() {
        for (Entry &e : entries_) e.val = kEmpty;
        size_ = used_ = 0;
    }
    size_t size() const { return size_; }

private:
    static constexpr int32_t kEmpty = INT32_MIN, kTomb = INT32_MIN + 1;
    struct Entry {
        int32_t key, val;
    };
    size_t slotOf(int32_t key) const { return ((uint32_t) key * 2654435761u) & mask_; }
    void rehash(size_t cap) {
        std::vector<Entry> old(std::move(entries_));
        entries_.assign(cap, Entry{0, kEmpty});
        mask_ = cap - 1;
        size_ = used_ = 0;
        for (const Entry &e : old)
            if (e.val != kEmpty && e.val != kTomb) set(e.key, e.val);
    }
    std::vector<Entry> entries_;
    size_t mask_ = 0, size_ = 0, used_ = 0;
};

}  // namespace cfa
