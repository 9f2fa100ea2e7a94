// Open-addressing int32 -> int32 hash map (linear probing, tombstones, power-of-two capacity).
// The spawner inserts one entry per spawned vehicle and never shrinks, so at city scale the table is tens of MiB and
// every access is a cache miss: key and value share one 8-byte entry (one miss per access, state folded into the value)
// and prefetch() lets the caller overlap the misses of a whole step.
// Growth never stalls the caller (a long run must not have a next_step() that takes milliseconds, tests/test_steady_state.py):
//   * a new table is calloc'ed — its pages cost nothing until they are touched — so an empty slot is all zero bits: values are
//     kept + 3 (the spawner stores vehicle numbers and -1; 0 = empty, 1 = tombstone);
//   * the old table is drained a few slots per operation, lookups look at both meanwhile.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace cfa {

class FlatMapI32 {
public:
    FlatMapI32() { cur_.init(1024); }

    // values must be >= -1
    bool lookup(int32_t key, int32_t &value) {
        migrateSome();
        if (const Entry *e = cur_.find(key)) {
            value = e->enc - kBias;
            return true;
        }
        if (draining())
            if (const Entry *e = old_.find(key)) {
                value = e->enc - kBias;
                return true;
            }
        return false;
    }
    bool contains(int32_t key) {
        int32_t v;
        return lookup(key, v);
    }
    void set(int32_t key, int32_t value) {
        migrateSome();
        if (draining()) {  // the key lives in at most one table: move it over
            if (Entry *e = old_.find(key)) old_.eraseAt(e);
        } else if ((cur_.used + 1) * 10 > (cur_.mask + 1) * 6) {
            startGrow((cur_.mask + 1) * 2);
        }
        cur_.set(key, value + kBias);
    }
    void erase(int32_t key) {
        migrateSome();
        if (Entry *e = cur_.find(key)) cur_.eraseAt(e);
        else if (draining())
            if (Entry *w = old_.find(key)) old_.eraseAt(w);
    }
    // make room for n entries without a growth on the way
    void reserve(size_t n) {
        finishMigration();
        size_t cap = cur_.mask + 1;
        while (n * 10 > cap * 6) cap *= 2;
        if (cap != cur_.mask + 1) {
            startGrow(cap);
            finishMigration();
        }
    }
    void prefetch(int32_t key) const {
        __builtin_prefetch(&cur_.slots.p[cur_.slotOf(key)], 1, 1);
        if (draining()) __builtin_prefetch(&old_.slots.p[old_.slotOf(key)], 1, 1);
    }
    void clear() {
        old_ = Table();
        drainAt_ = 0;
        cur_.init(cur_.mask + 1);
    }
    size_t size() const { return cur_.size + old_.size; }

private:
    static constexpr int32_t kBias = 3, kEmpty = 0, kTomb = 1;
    static constexpr size_t kMigratePerOp = 32;  // old-table slots visited per operation while a migration is under way
    struct Entry {
        int32_t key, enc;  // enc = value + kBias; 0 empty, 1 tombstone
    };
    struct Slots {  // calloc'ed array with value semantics (the spawner's state is copied for snapshots)
        Entry *p = nullptr;
        size_t n = 0;
        Slots() = default;
        Slots(const Slots &o) { *this = o; }
        Slots(Slots &&o) noexcept : p(o.p), n(o.n) {
            o.p = nullptr;
            o.n = 0;
        }
        Slots &operator=(const Slots &o) {
            if (this == &o) return *this;
            free(p);
            p = nullptr;
            n = o.n;
            if (n) {
                p = (Entry *) malloc(n * sizeof(Entry));
                memcpy(p, o.p, n * sizeof(Entry));
            }
            return *this;
        }
        Slots &operator=(Slots &&o) noexcept {
            if (this != &o) {
                free(p);
                p = o.p;
                n = o.n;
                o.p = nullptr;
                o.n = 0;
            }
            return *this;
        }
        ~Slots() { free(p); }
        void zeroed(size_t cap) {
            free(p);
            p = (Entry *) calloc(cap, sizeof(Entry));
            n = cap;
        }
    };
    struct Table {
        Slots slots;
        size_t mask = 0, size = 0, used = 0;
        void init(size_t cap) {
            slots.zeroed(cap);
            mask = cap - 1;
            size = used = 0;
        }
        size_t slotOf(int32_t key) const { return ((uint32_t) key * 2654435761u) & mask; }
        Entry *find(int32_t key) {
            if (!slots.n) return nullptr;
            size_t i = slotOf(key);
            for (;;) {
                Entry &e = slots.p[i];
                if (e.enc == kEmpty) return nullptr;
                if (e.enc != kTomb && e.key == key) return &e;
                i = (i + 1) & mask;
            }
        }
        void set(int32_t key, int32_t enc) {
            size_t i = slotOf(key), firstTomb = (size_t) -1;
            for (;;) {
                Entry &e = slots.p[i];
                if (e.enc == kEmpty) break;
                if (e.enc == kTomb) {
                    if (firstTomb == (size_t) -1) firstTomb = i;
                } else if (e.key == key) {
                    e.enc = enc;
                    return;
                }
                i = (i + 1) & mask;
            }
            if (firstTomb != (size_t) -1) i = firstTomb;
            else ++used;
            slots.p[i].key = key;
            slots.p[i].enc = enc;
            ++size;
        }
        void eraseAt(Entry *e) {
            e->enc = kTomb;
            --size;
        }
    };
    bool draining() const { return old_.slots.n != 0; }
    void startGrow(size_t cap) {
        finishMigration();  // (a second growth before the first has drained: cannot happen at this drain rate, kept safe)
        old_ = std::move(cur_);
        cur_ = Table();
        cur_.init(cap);
        drainAt_ = 0;
    }
    void migrateSome() {
        if (!draining()) return;
        const size_t end = std::min(drainAt_ + kMigratePerOp, old_.slots.n);
        for (; drainAt_ < end; ++drainAt_) {
            Entry &e = old_.slots.p[drainAt_];
            if (e.enc != kEmpty && e.enc != kTomb) {
                cur_.set(e.key, e.enc);
                e.enc = kTomb;
                --old_.size;
            }
        }
        if (drainAt_ == old_.slots.n) old_ = Table();
    }
    void finishMigration() {
        while (draining()) migrateSome();
    }
    Table cur_, old_;
    size_t drainAt_ = 0;
};

}  // namespace cfa
