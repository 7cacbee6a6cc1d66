// tests/sim/hip_emu/emu_runtime.cpp -- TEST AID (see hip/hip_runtime.h): fiber scheduler for emulated workgroups and a
// synchronous-enough stand-in for the few HIP runtime calls the host side of the kernels uses.
#include <execinfo.h>
#include <pthread.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "hip/hip_runtime.h"

thread_local emu_uint3 threadIdx, blockIdx, blockDim, gridDim;

// ---------------------------------------------------------------------------------------------- context switch (x86-64 SysV)
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch,.-emu_switch
)");
extern "C" void emu_switch(void** save_sp, void* load_sp);

namespace {

enum { ST_READY = 0, ST_BARRIER, ST_WAVE, ST_YIELD, ST_DONE };
constexpr size_t kStack = 96 * 1024;

struct Fiber {
    void* sp = nullptr;
    int state = ST_READY;
    unsigned gen = 0;  // barrier / wave generation the fiber waits to pass
};
struct Wave {
    int live = 0, arrived = 0;
    unsigned gen = 0;
    EmuWaveBuf buf;
    EmuI8Buf i8;
};
struct Group {  // one workgroup
    std::vector<Fiber> fib;
    std::vector<Wave> waves;
    char* stacks = nullptr;
    void* sched_sp = nullptr;
    int cur = -1, live = 0, bar_arrived = 0;
    unsigned bar_gen = 0;
    const std::function<void()>* body = nullptr;
    void* dyn_lds = nullptr;
    emu_uint3 bdim{}, bidx{}, gdim{};
};
thread_local Group* tl_group = nullptr;

void fiber_block(Group& g) { emu_switch(&g.fib[g.cur].sp, g.sched_sp); }

extern "C" void emu_fiber_main() {
    Group& g = *tl_group;
    (*g.body)();
    Fiber& f = g.fib[g.cur];
    f.state = ST_DONE;
    --g.live;
    Wave& w = g.waves[g.cur >> 6];
    --w.live;
    // a finished fiber no longer takes part in rendez-vous: release what only waited for it
    if (g.live > 0 && g.bar_arrived == g.live) {
        g.bar_arrived = 0;
        ++g.bar_gen;
    }
    if (w.live > 0 && w.arrived == w.live) {
        w.arrived = 0;
        ++w.gen;
    }
    fiber_block(g);
    abort();  // never resumed
}

// EMU_TRACE: a workgroup thread that stays inside one fiber for seconds (a loop that never reaches a rendez-vous) gets a
// signal from a watchdog and prints where it is
void on_usr1(int) {
    void* bt[32];
    const int n = backtrace(bt, 32);
    backtrace_symbols_fd(bt, n, 2);
}
std::atomic<long long> g_switches{0};

void run_group(Group& g, int order_mode, unsigned seed) {
    const int nt = (int)g.fib.size();
    tl_group = &g;
    static const bool trace0 = getenv("EMU_TRACE") != nullptr;
    std::atomic<bool> wd_stop{false};
    std::thread wd;
    if (trace0) {
        signal(SIGUSR1, on_usr1);
        pthread_t self = pthread_self();
        wd = std::thread([&wd_stop, self, &g]() {
            long long last = -1;
            for (int k = 0; !wd_stop; ++k) {
                usleep(100000);
                if (k % 30 != 29) continue;
                const long long now = g_switches.load();
                if (now == last) {
                    fprintf(stderr, "emu: wg %u fiber %d has not yielded for 3 s:\n", g.bidx.x, g.cur);
                    pthread_kill(self, SIGUSR1);
                }
                last = now;
            }
        });
    }
    blockIdx = g.bidx;
    blockDim = g.bdim;
    gridDim = g.gdim;
    g.stacks = (char*)mmap(nullptr, kStack * nt, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (g.stacks == (char*)MAP_FAILED) {
        perror("emu: mmap");
        abort();
    }
    for (int t = 0; t < nt; ++t) {
        char* top = g.stacks + kStack * (t + 1);
        void** sp = (void**)(((uintptr_t)top) & ~(uintptr_t)15);
        sp -= 2;
        sp[0] = (void*)&emu_fiber_main;  // ret target
        sp[1] = nullptr;                 // its (never used) return address slot
        sp -= 6;                         // rbp rbx r12-r15
        for (int i = 0; i < 6; ++i) sp[i] = nullptr;
        g.fib[t].sp = sp;
        g.fib[t].state = ST_READY;
    }
    g.live = nt;
    for (size_t w = 0; w < g.waves.size(); ++w) g.waves[w].live = std::min(64, nt - 64 * (int)w);
    std::vector<int> order(nt);
    for (int t = 0; t < nt; ++t) order[t] = order_mode == 1 ? nt - 1 - t : t;
    std::mt19937 rng(seed);
    const auto t_start = std::chrono::steady_clock::now();
    long long idle_passes = 0;
    static const bool trace = getenv("EMU_TRACE") != nullptr;
    auto t_trace = t_start;
    while (g.live > 0) {
        if (trace && std::chrono::steady_clock::now() - t_trace > std::chrono::seconds(3)) {
            t_trace = std::chrono::steady_clock::now();
            int hist[5] = {0};
            for (const Fiber& f : g.fib) hist[f.state]++;
            fprintf(stderr, "emu: wg %u live %d ready %d barrier %d (arrived %d) wave %d yield %d done %d |", g.bidx.x, g.live, hist[0],
                    hist[1], g.bar_arrived, hist[2], hist[3], hist[4]);
            for (size_t w = 0; w < g.waves.size(); ++w) fprintf(stderr, " w%zu %d/%d", w, g.waves[w].arrived, g.waves[w].live);
            fprintf(stderr, "\n");
        }
        if (order_mode == 2) std::shuffle(order.begin(), order.end(), rng);
        bool ran = false, polling = false;
        for (int k = 0; k < nt; ++k) {
            const int t = order[k];
            Fiber& f = g.fib[t];
            bool go = false;
            switch (f.state) {
                case ST_READY: go = true; break;
                case ST_BARRIER: go = f.gen != g.bar_gen; break;
                case ST_WAVE: go = f.gen != g.waves[t >> 6].gen; break;
                case ST_YIELD: go = true, polling = true; break;
                default: break;
            }
            if (!go) continue;
            if (f.state != ST_YIELD) ran = true;
            f.state = ST_READY;
            g.cur = t;
            threadIdx.x = (unsigned)t;
            threadIdx.y = threadIdx.z = 0;
            emu_switch(&g.sched_sp, f.sp);
            if (trace0) g_switches.fetch_add(1, std::memory_order_relaxed);
        }
        if (!ran) {
            if (!polling && g.live > 0) {
                fprintf(stderr, "emu: workgroup %u deadlocked (%d live fibers, %d at the barrier)\n", g.bidx.x, g.live, g.bar_arrived);
                abort();
            }
            // only pollers left: they wait for another workgroup
            if (++idle_passes % 64 == 0) {
                std::this_thread::yield();
                if (std::chrono::steady_clock::now() - t_start > std::chrono::seconds(600)) {
                    fprintf(stderr, "emu: workgroup %u still polling after 600 s\n", g.bidx.x);
                    abort();
                }
            }
        }
    }
    if (trace0) {
        wd_stop = true;
        wd.join();
    }
    munmap(g.stacks, kStack * nt);
    tl_group = nullptr;
}

}  // namespace

void emu_syncthreads() {
    Group& g = *tl_group;
    Fiber& f = g.fib[g.cur];
    f.gen = g.bar_gen;
    f.state = ST_BARRIER;
    if (++g.bar_arrived == g.live) {
        g.bar_arrived = 0;
        ++g.bar_gen;
    }
    fiber_block(g);
}
void emu_wave_sync() {
    Group& g = *tl_group;
    Fiber& f = g.fib[g.cur];
    Wave& w = g.waves[g.cur >> 6];
    f.gen = w.gen;
    f.state = ST_WAVE;
    if (++w.arrived == w.live) {
        w.arrived = 0;
        ++w.gen;
    }
    fiber_block(g);
}
void emu_yield() {
    Group& g = *tl_group;
    g.fib[g.cur].state = ST_YIELD;
    fiber_block(g);
}
long long emu_clock() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
EmuWaveBuf& emu_wave_buf() { return tl_group->waves[tl_group->cur >> 6].buf; }
EmuI8Buf& emu_wave_i8() { return tl_group->waves[tl_group->cur >> 6].i8; }
void* emu_dyn_lds() { return tl_group->dyn_lds; }

// ---------------------------------------------------------------------------------------------- streams and launches
struct EmuTask {
    std::mutex m;
    std::condition_variable cv;
    bool done = false;
    std::chrono::steady_clock::time_point t_done;
    void finish() {
        {
            std::lock_guard<std::mutex> lk(m);
            done = true;
            t_done = std::chrono::steady_clock::now();
        }
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
    }
    bool is_done() {
        std::lock_guard<std::mutex> lk(m);
        return done;
    }
};
struct EmuStream {
    std::mutex m;
    std::shared_ptr<EmuTask> tail;  // last launch queued on this stream
};
struct EmuEvent {
    std::shared_ptr<EmuTask> after;  // the launch the event was recorded behind (null: nothing pending)
    std::chrono::steady_clock::time_point t_record;
};
namespace {
EmuStream g_null_stream;
EmuStream* S(hipStream_t s) { return s ? s : &g_null_stream; }
std::shared_ptr<EmuTask> tail_of(hipStream_t s) {
    std::lock_guard<std::mutex> lk(S(s)->m);
    return S(s)->tail;
}
void drain(hipStream_t s) {
    if (auto t = tail_of(s)) t->wait();
}
}  // namespace

void emu_launch(std::function<void()> body, dim3 grid, dim3 block, size_t dyn_lds_bytes, hipStream_t stream) {
    auto task = std::make_shared<EmuTask>();
    std::shared_ptr<EmuTask> prev;
    {
        std::lock_guard<std::mutex> lk(S(stream)->m);
        prev = S(stream)->tail;
        S(stream)->tail = task;
    }
    int order_mode = 0;
    if (const char* e = getenv("EMU_ORDER")) order_mode = !strcmp(e, "reverse") ? 1 : (!strcmp(e, "shuffle") ? 2 : 0);
    std::thread([=]() {
        if (prev) prev->wait();  // stream order
        const int gx = (int)grid.x, gy = (int)grid.y, nb = gx * gy, nt = (int)block.x;  // (1-D blocks, 1-D / 2-D grids)
        // every workgroup of a grid of up to 1024 runs on its own OS thread at once (the solver's workgroups wait for each
        // other); larger grids -- extraction kernels on big images, whose workgroups never wait for one another -- are worked
        // off by 256 threads
        const int nthreads = nb <= 1024 ? nb : 256;
        std::atomic<int> next{0};
        std::vector<std::thread> th;
        th.reserve(nthreads);
        for (int w = 0; w < nthreads; ++w)
            th.emplace_back([&]() {
                for (int b = next.fetch_add(1); b < nb; b = next.fetch_add(1)) {
                    Group g;
                    g.fib.resize(nt);
                    g.waves.resize((nt + 63) / 64);
                    g.body = &body;
                    g.bdim = {(unsigned)nt, 1, 1};
                    g.bidx = {(unsigned)(b % gx), (unsigned)(b / gx), 0};
                    g.gdim = {(unsigned)gx, (unsigned)gy, 1};
                    static const int fill = getenv("EMU_LDS_FILL") ? (int)strtol(getenv("EMU_LDS_FILL"), nullptr, 0) : 0xcd;
                    std::vector<char> lds(dyn_lds_bytes + 64, (char)fill);  // LDS is not zero-initialised on the device either
                    g.dyn_lds = (void*)(((uintptr_t)lds.data() + 63) & ~(uintptr_t)63);
                    run_group(g, order_mode, 12345u + (unsigned)b);
                }
            });
        for (auto& t : th) t.join();
        task->finish();
    }).detach();
}

hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) {
    *n = 1;
    return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "emulated"; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) {
    int cus = 256;
    if (const char* e = getenv("EMU_CUS")) cus = atoi(e);
    *v = cus;
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    *s = new EmuStream();
    return hipSuccess;
}
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) {
    *s = new EmuStream();
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    drain(s);
    delete s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
    drain(s);
    return hipSuccess;
}
hipError_t hipStreamQuery(hipStream_t s) {
    auto t = tail_of(s);
    return (!t || t->is_done()) ? hipSuccess : hipErrorNotReady;
}
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) {
    if (e && e->after) e->after->wait();  // (host-blocking: stronger than the real call)
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) {
    *e = new EmuEvent();
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    e->after = tail_of(s);
    e->t_record = std::chrono::steady_clock::now();
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    if (e->after) e->after->wait();
    return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e) { return (!e->after || e->after->is_done()) ? hipSuccess : hipErrorNotReady; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    auto when = [](hipEvent_t e) { return (e->after && e->after->is_done() && e->after->t_done > e->t_record) ? e->after->t_done : e->t_record; };
    *ms = std::chrono::duration<float, std::milli>(when(b) - when(a)).count();
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t bytes) {
    *p = aligned_alloc(256, (bytes + 255) / 256 * 256);
    if (*p) memset(*p, 0xa5, bytes);  // fresh device memory holds garbage
    return *p ? hipSuccess : hipErrorInvalidValue;
}
hipError_t hipFree(void* p) {
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipMemsetAsync(void* p, int v, size_t bytes, hipStream_t s) {
    drain(s);
    memset(p, v, bytes);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t s) {
    drain(s);
    memcpy(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) {
    memcpy(dst, src, bytes);
    return hipSuccess;
}
