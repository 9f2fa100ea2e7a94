// TEST INFRASTRUCTURE ONLY (oracle/): makes the REFERENCE's lane change reproducible, without touching the reference.
//
// With laneChange=true the reference walks its lane-change candidates, and runs Engine::vehicleControl, in
// std::set<Vehicle*> order (engine.h:39-41, engine.cpp:374-390,402-413), i.e. by heap address.  Which address
// `new Vehicle` returns depends on everything the process allocated and freed before: the same
// configuration gives different results with a pipe instead of a terminal on stderr.  LD_PRELOADing this library into
// the process that runs oracle/_ref's cityflow_ref replaces the global operator new / delete: blocks of exactly
// CFX_VEHICLE_SIZE bytes (= sizeof(CityFlow::Vehicle), printed by _ref/probe_vehicle_size) come from a bump arena that
// never reuses memory, everything else goes to malloc.  Vehicle addresses then grow with creation order, which is the
// order include/cityflow_amd.h fixes for lane change — and the reference can be compared with the twin step for step.
#include <cstdio>
#include <cstdlib>
#include <new>
#include <sys/mman.h>

namespace {
char *g_base = nullptr, *g_next = nullptr, *g_end = nullptr;
size_t g_size = 0;
bool g_init = false;

void init() {
    g_init = true;
    const char *s = getenv("CFX_VEHICLE_SIZE");
    g_size = s ? (size_t) atol(s) : 0;
    if (!g_size) return;
    const size_t bytes = (size_t) 8 << 30;  // address space only; pages are committed when touched
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) {
        g_size = 0;
        return;
    }
    g_base = g_next = (char *) p;
    g_end = g_base + bytes;
}
inline void *take(size_t n) {
    if (!g_init) init();
    if (n == g_size && g_size && g_next + ((n + 15) & ~(size_t) 15) <= g_end) {  // single-threaded use (thread_num = 1)
        void *p = g_next;
        g_next += (n + 15) & ~(size_t) 15;
        return p;
    }
    void *p = malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
inline void give(void *p) {
    if (p >= (void *) g_base && p < (void *) g_end) return;  // never reused
    free(p);
}
}  // namespace

void *operator new(size_t n) { return take(n); }
void *operator new[](size_t n) { return take(n); }
void operator delete(void *p) noexcept { give(p); }
void operator delete[](void *p) noexcept { give(p); }
void operator delete(void *p, size_t) noexcept { give(p); }
void operator delete[](void *p, size_t) noexcept { give(p); }
