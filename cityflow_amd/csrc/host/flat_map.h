// Open-addressing int32 -> int32 hash map (linear probing, tombstones, power-of-two capacity).
// The spawner inserts one entry per spawned vehicle; a node-allocating std::unordered_map made that the most
// expensive host operation of a step.
#pragma once

#include <cstdint>
#include <vector>

namespace cfa {

class FlatMapI32 {
public:
    FlatMapI32() { rehash(1024); }

    // returns pointer to the value or nullptr
    int32_t *find(int32_t key) {
        size_t i = slotOf(key);
        for (;;) {
            uint8_t st = state_[i];
            if (st == kEmpty) return nullptr;
            if (st == kFull && keys_[i] == key) return &vals_[i];
            i = (i + 1) & mask_;
        }
    }
    void set(int32_t key, int32_t value) {
        if ((used_ + 1) * 10 > (mask_ + 1) * 6) rehash((mask_ + 1) * 2);
        size_t i = slotOf(key), firstTomb = (size_t) -1;
        for (;;) {
            uint8_t st = state_[i];
            if (st == kEmpty) break;
            if (st == kFull && keys_[i] == key) {
                vals_[i] = value;
                return;
            }
            if (st == kTomb && firstTomb == (size_t) -1) firstTomb = i;
            i = (i + 1) & mask_;
        }
        if (firstTomb != (size_t) -1) i = firstTomb;
        else ++used_;
        state_[i] = kFull;
        keys_[i] = key;
        vals_[i] = value;
        ++size_;
    }
    void erase(int32_t key) {
        size_t i = slotOf(key);
        for (;;) {
            uint8_t st = state_[i];
            if (st == kEmpty) return;
            if (st == kFull && keys_[i] == key) {
                state_[i] = kTomb;
                --size_;
                return;
            }
            i = (i + 1) & mask_;
        }
    }
    void clear() {
        std::fill(state_.begin(), state_.end(), (uint8_t) kEmpty);
        size_ = used_ = 0;
    }
    size_t size() const { return size_; }

private:
    enum : uint8_t { kEmpty = 0, kFull = 1, kTomb = 2 };
    size_t slotOf(int32_t key) const { return ((uint32_t) key * 2654435761u) & mask_; }
    void rehash(size_t cap) {
        std::vector<int32_t> ok(std::move(keys_)), ov(std::move(vals_));
        std::vector<uint8_t> os(std::move(state_));
        keys_.assign(cap, 0);
        vals_.assign(cap, 0);
        state_.assign(cap, (uint8_t) kEmpty);
        mask_ = cap - 1;
        size_ = used_ = 0;
        for (size_t i = 0; i < os.size(); ++i)
            if (os[i] == kFull) set(ok[i], ov[i]);
    }
    std::vector<int32_t> keys_, vals_;
    std::vector<uint8_t> state_;
    size_t mask_ = 0, size_ = 0, used_ = 0;
};

}  // namespace cfa
