// hipemu runtime: fiber scheduler.  TEST INFRASTRUCTURE ONLY (see hipemu.h).
#include "hipemu.h"

#include <algorithm>
#include <random>
#include <thread>

namespace hipemu {

thread_local Block* g_block = nullptr;
thread_local Fiber* g_fiber = nullptr;

static const size_t kStackBytes = 256 * 1024;

// Minimal x86-64 SysV context switch: save callee-saved registers on the current stack, publish the
// stack pointer, adopt the other one.  (glibc's swapcontext does a sigprocmask syscall per switch.)
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");

void yield_to_sched() {
    Fiber* f = g_fiber;
    hipemu_switch(&f->sp, g_block->sched_sp);
}

void block_barrier() {
    Block* b = g_block;
    Fiber* f = g_fiber;
    unsigned long gen = b->bar_gen;
    if (++b->bar_arrived >= b->alive) {
        b->bar_arrived = 0;
        b->bar_gen++;
        return;
    }
    f->wait = 1;
    f->wait_gen = gen;
    yield_to_sched();
}

void wave_sync() {
    Block* b = g_block;
    Fiber* f = g_fiber;
    Wave& w = b->waves[f->wave];
    unsigned long gen = w.gen;
    if (++w.arrived >= w.alive) {
        w.arrived = 0;
        w.gen++;
        return;
    }
    f->wait = 2;
    f->wait_gen = gen;
    yield_to_sched();
}

static void fiber_main() {
    Block* b = g_block;
    Fiber* f = g_fiber;
    b->entry(b->entry_arg);
    f->done = true;
    // exited threads no longer take part in barriers / collectives
    b->alive--;
    Wave& w = b->waves[f->wave];
    w.alive--;
    if (b->alive > 0 && b->bar_arrived >= b->alive && b->bar_arrived > 0) {
        b->bar_arrived = 0;
        b->bar_gen++;
    }
    if (w.alive > 0 && w.arrived >= w.alive && w.arrived > 0) {
        w.arrived = 0;
        w.gen++;
    }
    hipemu_switch(&f->sp, b->sched_sp);
    abort();  // a finished fiber is never resumed
}

struct Worker {
    std::vector<char*> stacks;
    std::vector<char> lds;
    ~Worker() {
        for (char* s : stacks) free(s);
    }
};

static int order_mode() {
    static int mode = [] {
        const char* e = getenv("HIPEMU_ORDER");
        if (!e) return 0;
        if (!strcmp(e, "rev")) return 1;
        if (!strcmp(e, "rand")) return 2;
        if (!strcmp(e, "wave")) return 3;      // greedy waves: a wave runs until every one of its fibers sits at a block barrier
        if (!strcmp(e, "waverev")) return 4;   // (or has finished) before the next wave gets a turn: one wave is a whole barrier
        if (!strcmp(e, "waverand")) return 5;  // interval ahead of the others, which exposes cross-wave LDS races between barriers
        return 0;
    }();
    return mode;
}

static void run_block(Worker& wk, dim3 bid, dim3 grid, dim3 bdim, size_t lds_bytes, void (*entry)(void*), void* arg) {
    int nthreads = (int)(bdim.x * bdim.y * bdim.z);
    int nwaves = (nthreads + 63) / 64;
    Block blk;
    blk.bid = bid;
    blk.bdim = bdim;
    blk.gdim = grid;
    blk.fibers.resize(nthreads);
    blk.waves.resize(nwaves);
    blk.alive = nthreads;
    blk.entry = entry;
    blk.entry_arg = arg;
    if (wk.lds.size() < lds_bytes + 64) wk.lds.resize(lds_bytes + 64);
    // poison LDS so that reads of never-written LDS are visible as NaNs / garbage
    memset(wk.lds.data(), 0xFF, wk.lds.size());
    uintptr_t base = (reinterpret_cast<uintptr_t>(wk.lds.data()) + 15) & ~uintptr_t(15);
    blk.lds = reinterpret_cast<char*>(base);
    blk.lds_bytes = lds_bytes;
    while ((int)wk.stacks.size() < nthreads) wk.stacks.push_back((char*)malloc(kStackBytes));
    g_block = &blk;
    for (int i = 0; i < nthreads; ++i) {
        Fiber& f = blk.fibers[i];
        f.linear = i;
        f.tid = dim3(i % bdim.x, (i / bdim.x) % bdim.y, i / (bdim.x * bdim.y));
        f.lane = i & 63;
        f.wave = i >> 6;
        blk.waves[f.wave].alive++;
        f.stack = wk.stacks[i];
        // initial frame: 6 callee-saved slots + return address (fiber_main); keep the ABI's 16-byte alignment
        uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStackBytes) & ~uintptr_t(15);
        void** a = reinterpret_cast<void**>(top - 16);
        a[0] = reinterpret_cast<void*>(&fiber_main);
        a[1] = nullptr;
        void** regs = a - 6;
        for (int r = 0; r < 6; ++r) regs[r] = nullptr;
        f.sp = regs;
    }
    std::vector<int> order(nthreads);
    for (int i = 0; i < nthreads; ++i) order[i] = i;
    int mode = order_mode();
    if (mode == 1) std::reverse(order.begin(), order.end());
    std::mt19937 rng(1234 + bid.x * 7919 + bid.y * 104729);
    int remaining = nthreads;
    if (mode >= 3) {
        std::vector<int> worder(nwaves);
        for (int i = 0; i < nwaves; ++i) worder[i] = i;
        if (mode == 4) std::reverse(worder.begin(), worder.end());
        while (remaining > 0) {
            if (mode == 5) std::shuffle(worder.begin(), worder.end(), rng);
            bool any = false;
            for (int wv : worder) {
                bool progress = true;
                while (progress) {
                    progress = false;
                    const int hi = std::min(nthreads, (wv + 1) * 64);
                    for (int idx = wv * 64; idx < hi; ++idx) {
                        Fiber& f = blk.fibers[idx];
                        if (f.done) continue;
                        if (f.wait == 1 && blk.bar_gen == f.wait_gen) continue;
                        if (f.wait == 2 && blk.waves[f.wave].gen == f.wait_gen) continue;
                        f.wait = 0;
                        progress = any = true;
                        g_fiber = &f;
                        hipemu_switch(&blk.sched_sp, f.sp);
                        if (f.done) remaining--;
                    }
                }
            }
            if (!any) {
                fprintf(stderr, "hipemu: DEADLOCK in block (%u,%u,%u): %d threads stuck (divergent barrier / collective?)\n", bid.x, bid.y,
                        bid.z, remaining);
                abort();
            }
        }
        g_block = nullptr;
        g_fiber = nullptr;
        return;
    }
    while (remaining > 0) {
        if (mode == 2) std::shuffle(order.begin(), order.end(), rng);
        bool progress = false;
        for (int idx : order) {
            Fiber& f = blk.fibers[idx];
            if (f.done) continue;
            if (f.wait == 1 && blk.bar_gen == f.wait_gen) continue;
            if (f.wait == 2 && blk.waves[f.wave].gen == f.wait_gen) continue;
            f.wait = 0;
            progress = true;
            g_fiber = &f;
            hipemu_switch(&blk.sched_sp, f.sp);
            if (f.done) remaining--;
        }
        if (!progress) {
            fprintf(stderr, "hipemu: DEADLOCK in block (%u,%u,%u): %d threads stuck (divergent barrier / collective?)\n", bid.x, bid.y,
                    bid.z, remaining);
            abort();
        }
    }
    g_block = nullptr;
    g_fiber = nullptr;
}

void run_grid(dim3 grid, dim3 block, size_t lds_bytes, void (*entry)(void*), void* arg) {
    size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    int nthr = 8;
    if (const char* e = getenv("HIPEMU_THREADS")) nthr = std::max(1, atoi(e));
    nthr = (int)std::min<size_t>(nthr, nblocks);
    auto work = [&](int w) {
        Worker wk;
        for (size_t b = w; b < nblocks; b += nthr) {
            dim3 bid((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y)));
            run_block(wk, bid, grid, block, lds_bytes, entry, arg);
        }
    };
    if (nthr <= 1) {
        work(0);
        return;
    }
    std::vector<std::thread> ts;
    for (int w = 0; w < nthr; ++w) ts.emplace_back(work, w);
    for (auto& t : ts) t.join();
}

}  // namespace hipemu
