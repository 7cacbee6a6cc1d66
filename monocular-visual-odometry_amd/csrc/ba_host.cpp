// csrc/ba_host.cpp -- host side of the bundle adjustment: what optimization::bundleAdjustment does before and after
// optimizer.optimize(50) (reference src/optimization/g2o_ba.cpp:193-271 graph construction, :298-316 read-back),
// turned into a flat, landmark-partitioned window that k_ba_lm (ba_kernels.hip) consumes:
//   plan      -- active edges, G landmark ranges balanced by edge count, edges sorted by (range, pose), adjacency tables
//   workspace -- pooled device + pinned memory per ctx (grow-only): a window is (re)built per frame with no hipMalloc,
//                no hipHostMalloc and ONE host-to-device copy; results come back through pinned mirrors the kernel
//                writes itself
//   service   -- one launch thread per device: the workgroups of a window meet at in-kernel hand-offs, so two BA grids
//                must never be half-resident at the same time; every BA launch of the process goes through this thread,
//                which batches up to BA_MAX_BATCH pending windows (of any ctx) into one grid.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "ba_types.h"
#include "mvo_internal.h"

int ba_kernel_set_lds_limit();
hipError_t ba_kernel_launch(const BaBatch& batch, int max_wgs, size_t lds_bytes, hipStream_t stream, int profile, int nr, int slots);
int ba_solver_class(int n);
hipError_t ba_service_launch(const BaServiceArgs& a, hipStream_t stream);

// hipFree / hipHostFree synchronise with every stream of the device, the never-ending resident solver grid included: the
// grid is taken off the device first (and kept off until the free is done).  Used by every grow / release path of the library.
void ba_service_free(int device, void* p, bool host);

int g_ba_use_mfma = 1;  // debug knobs (mvo_debug_set)
int g_ba_wgs = 0;       // 0 = automatic
int g_ba_same_l2 = 1;   // 0 = always write-through hand-offs
int g_ba_profile = 0;   // 1 = launch the instrumented kernel (per-phase cycle counters)
int g_ba_cu_share = 0;   // CUs a solver grid may take (0 = all)
int g_ba_xcd_reserve = 4;  // CUs per XCD a window leaves to other kernels
int g_ba_service = 1;     // throughput-mode windows of the 5-pose class and the resident solver service: 0 = never, 1 = while the
                          // offered load fills most of its slots (BaService::wanted), 2 = always
int g_ba_edge_rows = -1;   // -1 = automatic, 0 = Jacobian rows in LDS only (or fail), 1 = first 512 edges of a range in registers
int g_ba_uv_global = std::getenv("MVO_BA_UV_GLOBAL") ? std::atoi(std::getenv("MVO_BA_UV_GLOBAL")) : 1;     // 0 = measurements always in LDS (the form before the second half of round 3)
int g_ba_chunk_pieces = 0;  // 1 = one column piece per chunk when the Schur operands take several chunks (the round-2 form)
int g_ba_block_solver = 0;  // 1 = windows of <= 5 free poses use the workgroup-wide block LDL^T too
int g_ba_groups = std::getenv("MVO_BA_GROUPS") ? std::atoi(std::getenv("MVO_BA_GROUPS")) : 1;  // 0 = one flat Schur exchange whatever the window's size (A/B)
int g_ba_alias_sl = std::getenv("MVO_BA_ALIAS_SL") ? std::atoi(std::getenv("MVO_BA_ALIAS_SL")) : 1;  // 0 = the reduced system always has LDS of its own (A/B)

// The demand estimate of the resident solver service (a plain state machine over submission times, so that it can be
// replayed by the tests: mvo_debug_ba_demand_replay).
struct BaDemand {
    double t[128] = {0}, flip = -1e9, low_since = -1, last = 0, rate_avg = 0;  // rate_avg: submissions / s, exponentially averaged
    long long n = 0, total = 0, flips = 0;
    bool on = false;
    bool submit(double now);
};

namespace {

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) {
        size_t o = off;
        off = (off + bytes + 255) / 256 * 256;
        return o;
    }
};

// Everything the host computes for one window; offsets are relative to the start of the device block.
struct BaPlan {
    int F = 0, L = 0, E = 0, G = 1, nfree = 0, n = 0, NT = 1, npair = 1, nlow = 0, npk = 16, slice = 0, nsplit = 1, npar = 1, nseq = 1,
        ldu = 16, nhp = 1, maxEg = 0, maxLg = 0, max_dup = 0, fix_points = 0, e2_edges = 0, slots = 1, npt = 1, panel = 0, uv_global = 0,
        alias_sl = 0, groups = 1;
    bool service = false;  // solved by the resident solver service (inputs are read from the pinned image: no upload)
    size_t uarea = 0;
    size_t lds = 0;
    std::vector<int> wg_pt;
    // device block layout
    size_t o_desc = 0, o_pin = 0, o_ptsin = 0, o_wpt = 0, o_wed = 0, o_wps = 0, o_ep = 0, o_el = 0, o_uv = 0, o_ptstart = 0,
           o_ptl = 0, o_eof = 0, o_dup = 0, o_slot = 0, o_sp = 0, o_pkt = 0, upload_end = 0, x_end = 0;
    size_t o_uvd = 0, o_dxl = 0, o_stats = 0, o_pout = 0, o_pts = 0, o_xp = 0, o_xr = 0, o_xh = 0, o_xc = 0, total = 0;
    // pinned mirror layout (behind the upload staging)
    size_t m_stats = 0, m_poses = 0, m_pts = 0, m_trace = 0, pin_total = 0;
    bool runnable = true;  // false: nothing to optimise (no free vertex)
};

struct BaWorkspace {
    int device = 0;
    char* dev = nullptr;
    size_t dev_cap = 0;
    char* pin = nullptr;
    size_t pin_cap = 0;
    unsigned seq = 0;  // launch sequence number (tags)
    size_t x_sig[5] = {0, 0, 0, 0, 0};  // layout of the exchange areas the device block was last used with
    hipEvent_t ready = nullptr;
    BaPlan plan;
    bool in_flight = false;
    bool poisoned = false;  // a resident grid that never left may still write into these blocks: never reuse or free them
};

int ws_reserve(mvo_ctx* ctx, BaWorkspace& ws, size_t dev_bytes, size_t pin_bytes) {
    ws.device = ctx->device;
    if (ws.poisoned) {  // (leaked on purpose, see BaService::stop_resident)
        ws.dev = ws.pin = nullptr;
        ws.dev_cap = ws.pin_cap = 0;
        ws.seq = 0;
        std::memset(ws.x_sig, 0, sizeof ws.x_sig);
        ws.poisoned = false;
    }
    if (!ws.ready) MVO_HIP(hipEventCreateWithFlags(&ws.ready, hipEventDisableTiming));
    if (dev_bytes > ws.dev_cap) {
        MVO_HIP(hipStreamSynchronize(ctx->stream));
        if (ws.dev) ba_service_free(ctx->device, ws.dev, false);
        ws.dev = nullptr;
        ws.dev_cap = 0;
        std::memset(ws.x_sig, 0, sizeof ws.x_sig);  // fresh memory: the exchange areas count as laid out anew (cleared + ordered below)
        const size_t cap = dev_bytes + dev_bytes / 4 + 65536;
        MVO_HIP(hipMalloc((void**)&ws.dev, cap));
        ws.dev_cap = cap;
        // exchange granules are matched by tag: fresh memory must not hold a plausible one
        MVO_HIP(hipMemsetAsync(ws.dev, 0, cap, ctx->stream));
        ws.seq = 0;
    }
    if (pin_bytes > ws.pin_cap) {
        MVO_HIP(hipStreamSynchronize(ctx->stream));
        if (ws.pin) ba_service_free(ctx->device, ws.pin, true);
        ws.pin = nullptr;
        ws.pin_cap = 0;
        const size_t cap = pin_bytes + pin_bytes / 4 + 65536;
        MVO_HIP(hipHostMalloc((void**)&ws.pin, cap, hipHostMallocDefault));
        ws.pin_cap = cap;
    }
    return MVO_OK;
}
void ws_free(BaWorkspace& ws) {
    if (ws.dev && !ws.poisoned) ba_service_free(ws.device, ws.dev, false);
    if (ws.pin && !ws.poisoned) ba_service_free(ws.device, ws.pin, true);
    if (ws.ready) (void)hipEventDestroy(ws.ready);
    ws = BaWorkspace();
}

// ------------------------------------------------------------------------------------------------ launch service
struct BaJob {
    BaWorkspace* ws = nullptr;
    int use_mfma = 1;
    bool done = false;
    hipError_t err = hipSuccess;
    float ms = 0;   // duration of the launch this window was part of
    int batch = 0;  // windows in that launch
    int slot = -1;  // resident service: the slot the job was posted to (-1: not posted / launch path)
    unsigned long long seq = 0;  // and its sequence number there
};
// One launch in flight: its windows and the events around it (the completion thread turns it into results).
struct BaFlight {
    BaJob* jobs[BA_MAX_BATCH];
    int nj = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t launch_err = hipSuccess;
};
struct BaService {
    int device = 0;
    std::mutex m;
    std::condition_variable cv_work, cv_done, cv_flight;
    std::deque<BaJob*> q;
    std::deque<BaFlight*> flights;  // issued, not yet completed (at most BA_MAX_FLIGHTS)
    std::vector<BaFlight*> flight_pool;
    std::thread th, th_done;
    bool started = false;
    // resident solver service (k_ba_service): BA_SERVICE_SLOTS slots of `wgs_per_slot` workgroups stay on the device and
    // pull windows from their mailboxes; the launch thread only assigns slots
    bool resident = false;               // the grid is on the device
    std::atomic<bool> wedged{false};     // a grid of this service never left the device (stop_resident): no further grid is started, and
                                         // wanted() says no from then on -- windows take the launch path instead of failing at start_resident()
    BaMail* mail = nullptr;              // pinned host memory
    ba_u64* d_cmd = nullptr;             // device memory: 8 words per slot + one arrival counter per slot
    hipStream_t resident_stream = nullptr;
    unsigned long long slot_seq[BA_SERVICE_SLOTS] = {0};
    BaJob* slot_job[BA_SERVICE_SLOTS] = {nullptr};
    int slots_busy = 0;
    std::atomic<int> wgs_per_slot{14};  // workgroups per slot the NEXT grid gets (raised by planners, read by the scheduler)
    int wgs_launched = 0;               // what the grid on the device was launched with (scheduler thread only)
    int next_slot = 0;                  // slot assignment rotates: no slot sits idle for long while others work
    unsigned long long beat = 0;        // heartbeat written to every mailbox while the grid is resident (scheduler thread only)
    std::chrono::steady_clock::time_point beat_time{};
    std::atomic<int> park_forced{0};    // a caller needs the grid off the device now (memory is about to be freed)
    int free_gate = 0;                  // > 0: frees in progress, the grid must not come up (service mutex)
    long long path_switches = 0;        // times a launch-path window made the resident grid leave (mixed workloads)
    void heartbeat();
    std::atomic<int> q_pending{0};  // queued jobs (lets the scheduler poll without the mutex)
    bool park_requested = false;    // mvo_synchronize: take an idle resident grid off the device now
    std::condition_variable cv_slot;
    long long resident_jobs = 0, resident_starts = 0;
    double resident_cycles = 0;  // shader cycles of the windows the resident grid solved (sum; with `ms`: the clock under load)
    // Offered load of service-class windows = submission rate (over the last 32) x nominal solve time, in slots.  The
    // resident grid holds 2 x 14 CUs of every XCD whether its slots have work or not: it only pays while most slots are
    // busy (24 sequences x [extraction + BA]: ~14 of 16); with less demand (tracking rows in the loop, few sequences) the
    // windows take the launch path with the latency cut and the CUs go to whoever has work.  Coming: 128 submissions in a row
    // at a rate that fills 10 slots; going: the 40-ms average below 8 slots for 80 ms in a row (BaService::wanted).
    std::mutex m_demand;
    BaDemand demand;
    bool wanted();
    int start_resident();
    bool reap_locked();
    void stop_resident(std::unique_lock<std::mutex>& lk);
    // who submitted recently (workspace -> time of its last job): with several clients active a launch waits a moment for
    // a full batch, a lone client is never held back
    std::map<const BaWorkspace*, std::chrono::steady_clock::time_point> seen;
    // statistics (mvo_ba_launch_stats)
    long long launches = 0, windows = 0;
    double ms = 0;
    double t_idle = 0, t_batch = 0, t_launch = 0, t_sync = 0, t_post = 0;  // wall-clock of the loops' stages, ms
    void run();
    void complete();
};
#define BA_MAX_FLIGHTS 2
// heap-allocated and never destroyed: the detached service threads wait on these condition variables until the
// process ends (destroying a condition variable with a waiter would block exit)
BaService* g_service[16] = {nullptr};
std::mutex* g_service_start = new std::mutex();

// Launch thread: batches staged windows into grids and ISSUES them -- the next grid is queued behind the running one on
// the same stream (it starts when its predecessor has left the CUs; the workgroups of a window need each other
// resident, two grids must never be half-resident side by side).  The planner leaves a few CUs of every XCD to the
// callers' extraction / matching kernels (ba_stage), so those run WHILE a solve is in progress; with whole XCDs taken
// they only ran in the gaps between launches.  The completion thread waits for the launches in order and publishes
// their results.
void BaService::run() {
    (void)hipSetDevice(device);
    if (const char* e = std::getenv("MVO_BA_CU_SHARE")) g_ba_cu_share = std::atoi(e);        // development knobs, read once
    if (const char* e = std::getenv("MVO_BA_BLOCK_SOLVER")) g_ba_block_solver = std::atoi(e);
    hipStream_t stream = nullptr;
    (void)hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    (void)ba_kernel_set_lds_limit();
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) cus = 256;
    // stage clocks of one loop iteration: kept in locals and folded into the shared statistics under the lock
    auto tp = std::chrono::steady_clock::now();
    auto lap = [&tp](double& acc) {
        const auto now = std::chrono::steady_clock::now();
        acc += std::chrono::duration<double, std::milli>(now - tp).count();
        tp = now;
    };
    double l_idle = 0, l_batch = 0, l_launch = 0;
    for (;;) {
        BaFlight* fl = nullptr;
        {
            std::unique_lock<std::mutex> lk(m);
            // With the resident grid on the device this thread is its scheduler: it polls the completion words of the busy slots
            // (pinned host memory the device writes), hands finished windows back and posts queued windows to free slots.  A
            // slot is held exactly as long as the device works on it, not until the client comes back for the result.
            if (resident) (void)reap_locked();
            if (resident) heartbeat();
            if (park_forced.load(std::memory_order_acquire)) {  // (ba_service_with_grid_parked: hipFree would wait for the grid)
                if (resident) stop_resident(lk);
                park_forced.store(0, std::memory_order_release);
                cv_done.notify_all();
            }
            while (!q.empty() && q.front()->ws->plan.service) {
                // ---- resident solver service: no launch per window -- the job goes to a free slot of the resident grid
                if (!flights.empty()) {  // (launch-path grids and the resident grid never share the device)
                    cv_flight.wait(lk, [&] { return flights.empty(); });
                }
                // (a window planned for more workgroups per slot than the grid on the device has: the grid is relaunched)
                if (resident && q.front()->ws->plan.G > wgs_launched) stop_resident(lk);
                // frees in progress keep the grid off the device.  Checked AFTER the stop above (stop_resident drops the mutex
                // while it waits for the grid: a free may have begun meanwhile) and again after every wake-up.
                while (!resident && free_gate > 0) cv_work.wait(lk, [&] { return free_gate == 0; });
                if (!resident && start_resident() != 0) {
                    BaJob* j = q.front();
                    q.pop_front();
                    q_pending.fetch_sub(1, std::memory_order_relaxed);
                    j->err = hipErrorUnknown;
                    j->done = true;
                    cv_done.notify_all();
                    continue;
                }
                if (slots_busy >= BA_SERVICE_SLOTS) break;
                BaJob* j = q.front();
                q.pop_front();
                q_pending.fetch_sub(1, std::memory_order_relaxed);
                int sl = next_slot;
                while (slot_job[sl]) sl = (sl + 1) % BA_SERVICE_SLOTS;
                next_slot = (sl + 1) % BA_SERVICE_SLOTS;
                slot_job[sl] = j;
                ++slots_busy;
                ++resident_jobs;
                j->seq = ++slot_seq[sl];
                BaMail* mb = mail + sl;
                mb->desc = (ba_u64)(uintptr_t)(j->ws->pin + j->ws->plan.o_desc);
                mb->flags = (ba_u64)(j->ws->seq << 12) | ((ba_u64)(j->use_mfma ? 1 : 0) << 32) | ((ba_u64)(g_ba_same_l2 ? 1 : 0) << 33);
                __atomic_store_n(&mb->seq, j->seq, __ATOMIC_RELEASE);  // (fields first, the sequence number last)
                j->slot = sl;
            }
            if (resident && (q.empty() || q.front()->ws->plan.service)) {
                // nothing for the launch path: keep polling while slots are busy; an idle grid is taken off after 3 ms
                if (slots_busy > 0) {
                    // poll WITHOUT the mutex (this thread is the only writer of the slot table): the clients need the mutex
                    // to submit and to wake up -- a scheduler that spins on it starves them
                    lk.unlock();
                    for (;;) {
                        bool work = (q_pending.load(std::memory_order_acquire) > 0 && slots_busy < BA_SERVICE_SLOTS) ||
                                    park_forced.load(std::memory_order_acquire) != 0;
                        for (int sl = 0; sl < BA_SERVICE_SLOTS && !work; ++sl)
                            work = slot_job[sl] && __atomic_load_n(&mail[sl].done_seq, __ATOMIC_ACQUIRE) >= slot_job[sl]->seq;
                        if (work) break;
                        heartbeat();
                        for (int k = 0; k < 64; ++k) __builtin_ia32_pause();
                    }
                    continue;
                }
                // (a device-wide synchronisation of the caller waits for the grid as well: it must not linger)
                if (q.empty() && !cv_work.wait_for(lk, std::chrono::milliseconds(3), [&] { return !q.empty() || park_requested || park_forced.load() != 0; })) stop_resident(lk);
                else if (park_requested && q.empty() && slots_busy == 0) stop_resident(lk);
                park_requested = false;
                continue;
            }
            if (q.empty()) {
                cv_work.wait(lk, [&] { return !q.empty(); });
                continue;
            }
            lap(l_idle);
            if (resident) {  // a launch-path window: the resident grid leaves first
                ++path_switches;
                stop_resident(lk);
            }
            int share = g_ba_cu_share > 0 ? g_ba_cu_share : cus;
            share = std::max(1, std::min(share, cus));
            {   // batching: clients that submitted during the last 10 ms are expected back within a fraction of a solve
                const auto now = std::chrono::steady_clock::now();
                for (auto it = seen.begin(); it != seen.end();)
                    it = (now - it->second > std::chrono::milliseconds(10)) ? seen.erase(it) : std::next(it);
                const size_t fit = (size_t)std::max(1, share / std::max(1, q.front()->ws->plan.G));  // windows per grid
                const size_t want = std::min<size_t>(std::min<size_t>(BA_MAX_BATCH, fit), seen.size());
                // While a grid is still running there is no hurry: the next one is only worth queueing when it is full
                // (a partial grid behind a running one would hold back the windows that arrive a moment later for a whole
                // solve).  Once the device has no solver grid left, a partial batch waits a fraction of a solve at most.
                auto full = [&] { return q.size() >= want; };
                if (!full() && !flights.empty()) cv_work.wait(lk, [&] { return full() || flights.empty(); });
                if (!full()) cv_work.wait_for(lk, std::chrono::microseconds(250), full);
            }
            cv_flight.wait(lk, [&] { return flights.size() < BA_MAX_FLIGHTS; });
            if (flight_pool.empty()) {
                fl = new BaFlight();
                (void)hipEventCreate(&fl->e0);
                (void)hipEventCreate(&fl->e1);
            } else {
                fl = flight_pool.back();
                flight_pool.pop_back();
            }
            fl->nj = 0;
            // one workgroup per CU (the LDS slice is > 80 KB): a launch may not hold more workgroups than its share of
            // the CUs (a window larger than the share runs alone, on as many CUs as it needs)
            int sum_wgs = 0;
            while (fl->nj < BA_MAX_BATCH && !q.empty()) {
                const int G = q.front()->ws->plan.G;
                if (fl->nj && sum_wgs + G > share) break;
                if (fl->nj && ba_solver_class(q.front()->ws->plan.n) != ba_solver_class(fl->jobs[0]->ws->plan.n)) break;
                if (fl->nj && q.front()->ws->plan.slots != fl->jobs[0]->ws->plan.slots) break;  // (one kernel flavour per grid)
                sum_wgs += G;
                fl->jobs[fl->nj++] = q.front();
                q.pop_front();
                q_pending.fetch_sub(1, std::memory_order_relaxed);
            }
        }
        lap(l_batch);
        const int nj = fl->nj;
        BaBatch b{};
        b.nwin = nj;
        int maxG = 1, maxK = 1, slots = 0;
        size_t lds = 16;
        for (int i = 0; i < nj; ++i) {
            BaWorkspace* ws = fl->jobs[i]->ws;
            slots = ws->plan.slots;
            maxK = std::max(maxK, ws->plan.groups);
            (void)hipStreamWaitEvent(stream, ws->ready, 0);  // the window's upload (queued on its ctx stream)
            b.win[i] = (const BaDev*)(ws->dev + ws->plan.o_desc);
            b.tag_base[i] = ws->seq << 12;
            maxG = std::max(maxG, ws->plan.G);
            lds = std::max(lds, ws->plan.lds);
        }
        // window = block % stride: with the dispatcher's round-robin (block b on XCD b % 8) a stride of 8 or 16 keeps
        // every window's workgroups on ONE XCD (a window of > 32 workgroups does not fit an XCD's 32 CUs anyway)
        b.stride = cus >= 256 && maxG <= 16 ? 16 : (cus >= 256 && maxG <= 32 ? 8 : nj);
        if (cus >= 256 && maxK > 1) b.stride = (nj + 8 / maxK - 1) / (8 / maxK) * (8 / maxK);  // workgroup w of a window on XCD (w stride + window) mod 8: w mod K decides
        if (b.stride < nj) b.stride = nj;
        b.use_mfma = fl->jobs[0]->use_mfma;
        b.same_l2_ok = g_ba_same_l2;
        (void)hipEventRecord(fl->e0, stream);
        const int cls = ba_solver_class(fl->jobs[0]->ws->plan.n);
        fl->launch_err = ba_kernel_launch(b, maxG, lds, stream, g_ba_profile, cls == 32 && g_ba_block_solver ? -32 : cls, slots);
        (void)hipEventRecord(fl->e1, stream);
        lap(l_launch);
        {
            std::lock_guard<std::mutex> lk(m);
            flights.push_back(fl);
            t_idle += l_idle, t_batch += l_batch, t_launch += l_launch;
            l_idle = l_batch = l_launch = 0;
        }
        cv_flight.notify_all();
    }
}
}  // namespace
// Offered load of service-class windows -> should they go to the resident grid?  `now` in seconds (any monotonic origin).
bool BaDemand::submit(double now) {
    const double kSolve = 3.8e-3, kTau = 40e-3;  // nominal solve time of a window on a slot; time constant of the rate average
    const double dt = total ? now - last : 0.0;
    if (total && dt > 0.1) {
        // the callers paused (a barrier, a synchronisation, the end of a run): not low demand -- the decision stands, the
        // averages start over from where the decision would put them
        n = 0;
        rate_avg = on ? 0.5 * BA_SERVICE_SLOTS / kSolve : 0.0;
        low_since = -1;
    } else {
        if (dt > 0.03) n = 0;  // (the 64-submission window restarts after a long gap; sequences in lock step leave gaps of a step between their bursts -- those count)
        rate_avg *= std::exp(-dt / kTau);
    }
    const double load_avg = rate_avg * kSolve;  // slots kept busy, averaged over ~40 ms, as of just before this submission
    rate_avg += 1.0 / kTau;
    last = now;
    ++total;
    t[n++ & 127] = now;
    if (!on) {
        // coming: 128 submissions in a row at a rate that fills 8 of the 16 slots (the bursts of a slower loop never get there,
        // nor does a chance cluster of independent arrivals at half that rate).  The threshold has to lie well BELOW what the
        // launch path delivers when it is saturated (~2500 windows / s = 9.6 slots with 32 sequences in a closed loop): at 10
        // slots a run whose first steps were slow never saw the rate that would have brought the grid up, and stayed at half
        // the throughput for good (one of three driver-command runs on the same code, round 4).  128 in a row, not 64 (round 5):
        // independent arrivals at 1500 windows / s (5.7 slots: the tracking rows in the loop) show a 64-long cluster at the
        // 8-slot rate every second or so -- each one cost a grid start and a drain --, a 128-long one a few times a minute.
        if (n >= 128 && now - flip > 0.02) {
            const double span = now - t[n & 127];  // (the oldest of the 128)
            if (127.0 / std::max(span, 1e-6) * kSolve >= 0.5 * BA_SERVICE_SLOTS) {
                on = true;
                flip = now;
                low_since = -1;
                rate_avg = std::max(rate_avg, 0.5 * BA_SERVICE_SLOTS / kSolve);
                ++flips;
            }
        }
    } else if (load_avg > 0.4375 * BA_SERVICE_SLOTS) {
        low_since = -1;
    } else {
        // going: the average below 7 of the 16 slots for 80 ms in a row (the last steps of a run -- callers finishing one after
        // the other -- and the first ones after a pause look like low demand for a few milliseconds).  With the grid ON a loop
        // that waits for the solver offers far more than that (the grid delivers ~5000 windows/s = 19 slots' worth of
        // submissions); callers that offer less than 7 with the grid on are bound elsewhere (tracking rows in the loop: ~1500
        // windows/s = 5.7 slots, 1800 frames/s with the grid vs 2470 without) and a chance cluster of their arrivals that
        // brought the grid up must not keep it (round 4 left at 5: above which such a load stayed on for good).
        if (low_since < 0) low_since = now;
        if (now - low_since > 0.08) {
            on = false;
            flip = now;
            low_since = -1;
            ++flips;
        }
    }
    return on;
}
namespace {
bool BaService::wanted() {
    if (wedged.load(std::memory_order_relaxed)) return false;
    if (g_ba_service != 1) return g_ba_service == 2;
    std::lock_guard<std::mutex> lk(m_demand);
    return demand.submit(std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count());
}
// finished windows of the resident grid -> their clients (service mutex held); true if any slot was freed
bool BaService::reap_locked() {
    bool any = false;
    for (int sl = 0; sl < BA_SERVICE_SLOTS; ++sl) {
        BaJob* j = slot_job[sl];
        if (!j || __atomic_load_n(&mail[sl].done_seq, __ATOMIC_ACQUIRE) < j->seq) continue;
        const BaStatsDev* sd = (const BaStatsDev*)(j->ws->pin + j->ws->plan.m_stats);
        j->ms = (float)(sd->solve_ticks * 1e-5);  // 100 MHz ticks -> ms
        resident_cycles += (double)sd->phase[14];
        j->batch = 1;
        j->done = true;
        slot_job[sl] = nullptr;
        --slots_busy;
        ++launches;  // (statistics: every window of the resident grid counts as its own launch of `ms`)
        ++windows;
        ms += j->ms;
        any = true;
    }
    if (any) cv_done.notify_all();
    return any;
}
void BaService::heartbeat() {
    // (scheduler thread; the mailboxes are pinned host memory only this thread writes)
    const auto now = std::chrono::steady_clock::now();
    if (now - beat_time < std::chrono::milliseconds(200)) return;
    beat_time = now;
    ++beat;
    if (mail)
        for (int sl = 0; sl < BA_SERVICE_SLOTS; ++sl) __atomic_store_n(&mail[sl].beat, (ba_u64)beat, __ATOMIC_RELAXED);
}
int BaService::start_resident() {
    if (wedged) return -1;
    // (called with the service mutex held; everything here is quick)
    if (!mail) {
        if (hipHostMalloc((void**)&mail, sizeof(BaMail) * BA_SERVICE_SLOTS, hipHostMallocDefault) != hipSuccess) return -1;
        std::memset(mail, 0, sizeof(BaMail) * BA_SERVICE_SLOTS);
        if (hipMalloc((void**)&d_cmd, sizeof(ba_u64) * 9 * BA_SERVICE_SLOTS) != hipSuccess) return -1;
        // The resident grid never ends: nothing else may sit behind it in a hardware queue.  Streams of one priority share the
        // runtime's pool of hardware queues (GPU_MAX_HW_QUEUES) round-robin; a stream of its own priority class gets a
        // queue of its own.
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&resident_stream, hipStreamNonBlocking, greatest) != hipSuccess) return -1;
    }
    BaServiceArgs a{};
    a.mail = mail;
    a.cmd = d_cmd;
    a.arrived = d_cmd + 8 * BA_SERVICE_SLOTS;
    a.nslots = BA_SERVICE_SLOTS;
    a.wgs_per_slot = wgs_launched = wgs_per_slot.load(std::memory_order_relaxed);
    ba_u64 init[9 * BA_SERVICE_SLOTS] = {0};
    for (int sl = 0; sl < BA_SERVICE_SLOTS; ++sl) {
        a.first_seq[sl] = slot_seq[sl];
        init[8 * sl] = slot_seq[sl];  // (the republished sequence number starts where the mailbox stands: no job)
        mail[sl].seq = slot_seq[sl];
        mail[sl].stop = 0;
        mail[sl].done_seq = slot_seq[sl];
        mail[sl].beat = ++beat;
    }
    if (hipMemcpyAsync(d_cmd, init, sizeof(init), hipMemcpyHostToDevice, resident_stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(resident_stream) != hipSuccess) return -1;
    if (ba_service_launch(a, resident_stream) != hipSuccess) return -1;
    resident = true;
    ++resident_starts;
    return 0;
}
void BaService::stop_resident(std::unique_lock<std::mutex>& lk) {
    const auto t0 = std::chrono::steady_clock::now();
    while (slots_busy > 0) {  // (this thread is the one that reaps)
        if (!reap_locked()) {
            lk.unlock();
            std::this_thread::yield();
            lk.lock();
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) break;  // (a slot that never answers: give up waiting)
        }
    }
    // still outstanding after the bail-out: their clients must not wait for ever -- but they get their (failed) jobs back only
    // once the grid has left the device: until then it may still be writing into their workspaces
    BaJob* stuck[BA_SERVICE_SLOTS];
    int nstuck = 0;
    for (int sl = 0; sl < BA_SERVICE_SLOTS; ++sl)
        if (BaJob* j = slot_job[sl]) {
            stuck[nstuck++] = j;
            slot_job[sl] = nullptr;
            --slots_busy;
        }
    for (int sl = 0; sl < BA_SERVICE_SLOTS; ++sl) __atomic_store_n(&mail[sl].stop, (ba_u64)1, __ATOMIC_RELEASE);
    lk.unlock();
    // the grid leaves within microseconds of seeing `stop` -- unless a slot is truly wedged (then a plain synchronize would never
    // return and the stuck clients would wait for ever): poll with a deadline of its own
    bool left = false;
    const auto t1 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(resident_stream);
        if (q != hipErrorNotReady) {  // hipSuccess, or an error the stream will keep reporting: nothing left to wait for
            left = true;
            if (q != hipSuccess) (void)hipGetLastError();
            break;
        }
        if (std::chrono::steady_clock::now() - t1 > std::chrono::seconds(nstuck ? 10 : 30)) break;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    lk.lock();
    if (!left) {
        // the grid is still on the device and may still write into the workspaces of the jobs it never finished: those blocks
        // are leaked (never reused, never freed), the service stays down for this process (windows take the launch path)
        std::fprintf(stderr, "[mvo] resident solver grid did not leave within its deadline (%d unfinished window(s)): their workspaces are "
                             "abandoned, the service is disabled\n", nstuck);
        for (int i = 0; i < nstuck; ++i)
            if (stuck[i]->ws) stuck[i]->ws->poisoned = true;
        wedged = true;
    }
    for (int i = 0; i < nstuck; ++i) {
        stuck[i]->err = hipErrorUnknown;
        stuck[i]->done = true;
    }
    resident = false;
    cv_done.notify_all();
}
void BaService::complete() {
    (void)hipSetDevice(device);
    auto tp = std::chrono::steady_clock::now();
    auto lap = [&tp](double& acc) {
        const auto now = std::chrono::steady_clock::now();
        acc += std::chrono::duration<double, std::milli>(now - tp).count();
        tp = now;
    };
    for (;;) {
        BaFlight* fl;
        double l_wait = 0, l_sync = 0, l_post = 0;
        {
            std::unique_lock<std::mutex> lk(m);
            cv_flight.wait(lk, [&] { return !flights.empty(); });
            fl = flights.front();
        }
        lap(l_wait);
        hipError_t se = hipEventSynchronize(fl->e1);
        lap(l_sync);
        float t = 0;
        if (fl->launch_err == hipSuccess && se == hipSuccess) (void)hipEventElapsedTime(&t, fl->e0, fl->e1);
        else (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lk(m);
            ++launches;
            windows += fl->nj;
            ms += t;
            for (int i = 0; i < fl->nj; ++i) {
                fl->jobs[i]->err = fl->launch_err != hipSuccess ? fl->launch_err : se;
                fl->jobs[i]->ms = t;
                fl->jobs[i]->batch = fl->nj;
                fl->jobs[i]->done = true;
            }
            flights.pop_front();
            flight_pool.push_back(fl);
            lap(l_post);
            t_sync += l_sync, t_post += l_post;
        }
        cv_done.notify_all();
        cv_flight.notify_all();
        cv_work.notify_all();  // (the launch thread may be waiting for "no grid left")
    }
}
// at process exit a resident grid must not outlive the host: post `stop` to every slot and wait for the grid (running
// solves finish first; their results simply are not collected any more)
void ba_service_shutdown() {
    for (int d = 0; d < 16; ++d) {
        BaService* sp = g_service[d];
        if (!sp) continue;
        std::unique_lock<std::mutex> lk(sp->m);
        if (!sp->resident) continue;
        for (int sl = 0; sl < BA_SERVICE_SLOTS; ++sl) __atomic_store_n(&sp->mail[sl].stop, (ba_u64)1, __ATOMIC_RELEASE);
        lk.unlock();
        (void)hipSetDevice(sp->device);
        (void)hipStreamSynchronize(sp->resident_stream);
        lk.lock();
        sp->resident = false;
    }
}
BaService& service_for(int device) {
    std::lock_guard<std::mutex> lk(*g_service_start);
    if (!g_service[device & 15]) {
        g_service[device & 15] = new BaService();
        static bool hooked = false;
        if (!hooked) {
            hooked = true;
            std::atexit(ba_service_shutdown);
        }
    }
    BaService& s = *g_service[device & 15];
    if (!s.started) {
        s.device = device;
        s.started = true;
        if (const char* e = std::getenv("MVO_BA_SERVICE")) g_ba_service = std::atoi(e);
        s.th = std::thread([&s] { s.run(); });
        s.th.detach();  // lives as long as the process; blocks on its queue when idle
        s.th_done = std::thread([&s] { s.complete(); });
        s.th_done.detach();
    }
    return s;
}
}  // namespace
void ba_service_free(int device, void* p, bool host) {
    if (!p) return;
    BaService* sp = nullptr;
    if (device >= 0 && device < 16) {
        std::lock_guard<std::mutex> lk(*g_service_start);
        sp = g_service[device & 15];
    }
    bool gated = false;
    if (sp && sp->started) {
        std::unique_lock<std::mutex> lk(sp->m);
        ++sp->free_gate;
        gated = true;
        if (sp->resident) {
            sp->park_forced.store(1, std::memory_order_release);
            sp->cv_work.notify_all();
            // (running windows finish first: a few milliseconds; bounded in case the scheduler is stuck)
            sp->cv_done.wait_for(lk, std::chrono::seconds(2), [&] { return !sp->resident; });
        }
    }
    if (host) (void)hipHostFree(p);
    else (void)hipFree(p);
    if (gated) {
        {
            std::lock_guard<std::mutex> lk(sp->m);
            --sp->free_gate;
        }
        sp->cv_work.notify_all();
    }
}
namespace {
void service_submit(BaService& s, BaJob* jobs, int n) {
    {
        std::lock_guard<std::mutex> lk(s.m);
        const auto now = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) {
            s.q.push_back(&jobs[i]);
            s.seen[jobs[i].ws] = now;
        }
        s.q_pending.fetch_add(n, std::memory_order_release);
    }
    s.cv_work.notify_one();
}
void service_wait(BaService& s, BaJob* jobs, int n) {
    std::unique_lock<std::mutex> lk(s.m);
    s.cv_done.wait(lk, [&] {
        for (int i = 0; i < n; ++i)
            if (!jobs[i].done) return false;
        return true;
    });
}

// ------------------------------------------------------------------------------------------------ planning + staging
// Builds the plan of a window and writes the upload image (descriptor included) into the pinned staging buffer.
// MVO_HOST_TIMING=1: wall clock of the stages of ba_stage (printed every 200 windows to stderr; development aid)
struct StageTimes {
    double acc[4] = {0, 0, 0, 0};
    long n = 0;
    std::chrono::steady_clock::time_point t;
    bool on = std::getenv("MVO_HOST_TIMING") != nullptr;
    void start() {
        if (on) t = std::chrono::steady_clock::now();
    }
    void lap(int k) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        acc[k] += std::chrono::duration<double, std::micro>(now - t).count();
        t = now;
    }
    void done() {
        if (!on || ++n % 200) return;
        std::fprintf(stderr, "[mvo ba_stage us/window] plan %.1f tables %.1f reserve %.1f image %.1f\n", acc[0] / 200, acc[1] / 200,
                     acc[2] / 200, acc[3] / 200);
        acc[0] = acc[1] = acc[2] = acc[3] = 0;
    }
};
static thread_local StageTimes g_stage_times;

// Working vectors of ba_stage, kept per host thread: a BA5 window needs ~0.5 MB of them per call, and the larger ones
// (150 KB of sorted measurements) sit above malloc's mmap threshold -- fresh pages every frame otherwise.
struct StageScratch {
    std::vector<int> pose_slot, slot_pose, act, deg, wg_edge, owner, wg_pose, cnt, e_pose, e_point, ptstart, ptlist, cur, cur2;
    std::vector<double> e_uv;
    std::vector<short> eof, dup, seen, pkt;
    int pkt_n = -1;
};
static thread_local StageScratch g_stage_scratch;

int ba_stage(mvo_ctx* ctx, const mvo_ba_problem* p, BaWorkspace& ws, bool want_trace) {
    g_stage_times.start();
    StageScratch& SC = g_stage_scratch;
    BaPlan& P = ws.plan;
    P = BaPlan();
    const int F = p->n_poses, L = p->n_points;
    if (F > BA_MAX_POSES) return mvo_set_err(ctx, MVO_ERR_INVALID, "more than 20 poses in the window (vo.h kBuffSize_)", hipSuccess);
    // information matrix must be symmetric positive definite: Omega = Lc^T Lc
    const double a = p->info[0], b = p->info[1], c = p->info[2], d = p->info[3];
    if (!(a > 0) || std::fabs(b - c) > 1e-12 * (std::fabs(a) + std::fabs(d)) || !(a * d - b * b > 0))
        return mvo_set_err(ctx, MVO_ERR_INVALID, "information matrix must be symmetric positive definite", hipSuccess);
    const double lc00 = std::sqrt(a), lc01 = b / lc00, lc11 = std::sqrt(d - lc01 * lc01);
    std::vector<int>&pose_slot = SC.pose_slot, &slot_pose = SC.slot_pose;
    pose_slot.assign(F, -1);
    slot_pose.clear();
    for (int i = 0; i < F; ++i)
        if (!(p->pose_fixed && p->pose_fixed[i])) {
            pose_slot[i] = (int)slot_pose.size();
            slot_pose.push_back(i);
        }
    const int nfree = (int)slot_pose.size();
    // active edges: SparseOptimizer::initializeOptimization drops edges whose vertices are all fixed
    std::vector<int>&act = SC.act, &deg = SC.deg;
    act.clear();
    act.reserve(p->n_edges);
    deg.assign(L, 0);
    for (int e = 0; e < p->n_edges; ++e) {
        if (pose_slot[p->edge_pose[e]] < 0 && p->fix_points) continue;
        act.push_back(e);
        deg[p->edge_point[e]]++;
    }
    const int E = (int)act.size();
    const int n = 6 * nfree;
    const int NT = (n + 1 + 15) / 16, npair = NT * (NT + 1) / 2;
    const int ldu = 16 * NT, nhp = BA_HP * nfree + 1;
    // packed entries of the reduced system; the exchange rows also hold the pose-block partials that ride along
    const int nlow = n * (n + 1) / 2 + n, npk = std::max(16, (nlow + nhp + 15) & ~15);
    const bool do_schur = !p->fix_points && n > 0;
    P.F = F;
    P.L = L;
    P.E = E;
    P.nfree = nfree;
    P.n = n;
    P.NT = NT;
    P.npair = npair;
    P.nlow = nlow;
    P.npk = npk;
    P.ldu = ldu;
    P.nhp = nhp;
    P.fix_points = p->fix_points ? 1 : 0;
    P.runnable = !(F == 0 && (L == 0 || p->fix_points)) && (nfree > 0 || !p->fix_points);
    // ---- choose G and the landmark ranges (balanced by edge count).  A range's per-edge Jacobian rows live in the
    // registers of its 512 threads (BA_EDGE_SLOTS each), its landmark state and one chunk of the Schur operands in LDS.
    // The dispatcher places block b of a grid on XCD b % 8, whatever else runs there: a kernel of the caller's (extraction,
    // matching) cannot finish before it got CUs on EVERY XCD.  A solver workgroup owns its CU, so a window never takes a
    // whole XCD: `ba_xcd_reserve` CUs (default 4 of 32) stay free on each.  Latency mode (default): one window per XCD,
    // up to 28 workgroups for the 5-keyframe window of the benchmark (~330 edges each); throughput mode: two windows per
    // XCD, up to 14 (~670 edges each, 4 CUs of an XCD left free), half the CUs per window at some 40 % more time per solve.  Larger windows get more
    // workgroups: a range holds at most 1024 edges and must fit the LDS.
    static const bool plan_trace = std::getenv("MVO_BA_PLAN_TRACE") != nullptr;  // development aid: the planner's LDS fits
    static const int env_reserve = std::getenv("MVO_BA_XCD_RESERVE") ? std::atoi(std::getenv("MVO_BA_XCD_RESERVE")) : -1;
    // (measured with 24 sequences in flight: 2 x 14 workgroups per XCD leave the extraction kernels too little -- 3150
    // frames/s with the shards waiting for extraction --, 2 x 12 make the solves too slow -- 3550 --, 2 x 13 give 3800)
    // throughput mode = "many sequences share this GPU".  Windows the resident service can take (full BA, 5-pose class) go
    // there with the throughput cut while the demand keeps its slots busy; below that they take the launch path with the
    // latency cut (BaService::wanted).  Other windows of the mode keep the throughput cut on the launch path.
    bool throughput = ctx->ba_throughput_mode != 0;
    const bool svc_class = throughput && !p->fix_points && !g_ba_profile && !g_ba_block_solver && P.runnable && ba_solver_class(n) == 32;
    const bool svc = svc_class && !ctx->ba_never_resident && service_for(ctx->device).wanted();
    if (svc_class && !svc && g_ba_service == 1 && !ctx->ba_never_resident) throughput = false;
    // (round 5: the same reserve in both modes -- throughput mode = 2 x 14 workgroups per XCD.  Up to round 4 it left two CUs more
    // (2 x 13): the frame kernels of 32 sequences did not fit the 4 CUs per XCD that 2 x 14 leave -- 3150 frames/s in round 3, 5125 vs
    // 5063 in round 4; with the round-5 frame kernels: 5700 at 2 x 14 (solver slots 98 % busy, window 2.75 ms) vs 5060 at 2 x 13
    // (3.10 ms); 2 x 15 starves the extraction for good (1800))
    const int reserve = env_reserve >= 0 ? env_reserve : g_ba_xcd_reserve;
    const int per_xcd = std::max(8, 32 - std::max(0, std::min(reserve, 16)));
    const int g_cap = throughput ? per_xcd / 2 : per_xcd;
    if (throughput) {  // (monotonic: the grid is relaunched by its scheduler when a window needs more than it has)
        std::atomic<int>& wps = service_for(ctx->device).wgs_per_slot;
        int cur = wps.load(std::memory_order_relaxed);
        while (cur < per_xcd / 2 && !wps.compare_exchange_weak(cur, per_xcd / 2)) {
        }
    }
    int G = 1;
    while (G < g_cap && E > 160 * G) G = std::min(2 * G, g_cap);
    if (g_ba_wgs > 0) G = g_ba_wgs;
    if (const char* env = std::getenv("MVO_BA_WGS")) G = std::max(1, std::atoi(env));
    G = std::max(1, std::min(G, BA_MAX_WGS));
    int env_nsplit = 0;
    if (const char* env = std::getenv("MVO_BA_NSPLIT")) env_nsplit = std::max(1, std::atoi(env));
    std::vector<int>& wg_pt = P.wg_pt;
    std::vector<int>&wg_edge = SC.wg_edge, &owner = SC.owner, &wg_pose = SC.wg_pose;
    owner.assign(L, 0);
    int maxEg = 0, maxLg = 0, maxEpose = 0, nsplit = 1, npar = 1, nseq = 1;
    size_t uarea = 0;
    for (;;) {
        wg_pt.assign(G + 1, 0);
        wg_edge.assign(G + 1, 0);
        int l = 0, eacc = 0;
        for (int g = 0; g < G; ++g) {
            wg_pt[g] = l;
            wg_edge[g] = eacc;
            const long target = (long)E * (g + 1) / G;
            while (l < L && (g == G - 1 || eacc < target)) eacc += deg[l++];
        }
        wg_pt[G] = L;
        wg_edge[G] = E;
        maxEg = maxLg = 0;
        for (int g = 0; g < G; ++g) {
            maxEg = std::max(maxEg, wg_edge[g + 1] - wg_edge[g]);
            maxLg = std::max(maxLg, wg_pt[g + 1] - wg_pt[g]);
        }
        // edges per (range, pose): the pose-block chains stage whole poses
        for (int g = 0; g < G; ++g)
            for (int ll = wg_pt[g]; ll < wg_pt[g + 1]; ++ll) owner[ll] = g;
        wg_pose.assign((size_t)G * (F + 1), 0);
        {
            std::vector<int>& cnt = SC.cnt;
            cnt.assign((size_t)G * std::max(F, 1), 0);
            for (int e : act) cnt[(size_t)owner[p->edge_point[e]] * F + p->edge_pose[e]]++;
            int acc = 0;
            maxEpose = 0;
            for (int g = 0; g < G; ++g) {
                for (int q = 0; q < F; ++q) {
                    wg_pose[(size_t)g * (F + 1) + q] = acc;
                    acc += cnt[(size_t)g * F + q];
                    maxEpose = std::max(maxEpose, cnt[(size_t)g * F + q]);
                }
                wg_pose[(size_t)g * (F + 1) + F] = acc;
            }
        }
        // column pieces of the Schur chains: as many pieces side by side (on different waves) as there are idle waves, in
        // every chunk; as few chunks as the LDS budget allows (a wave keeps the running sums of <= 2 tile pairs)
        const int msteps = (3 * maxLg + 3) / 4;
        // where the Jacobian rows of the edges live: all in LDS when the range has the room (fewest registers: the
        // kernel flavour without register-resident rows), else the first 512 of a range in the registers of its threads
        bool fits = false;
        const int max_seq = (do_schur && npair <= 16) ? 8 : 1;
        for (int all_lds = (g_ba_edge_rows == 1 ? 0 : 1); all_lds >= (g_ba_edge_rows == 0 ? 1 : 0) && !fits; --all_lds) {
            P.e2_edges = all_lds ? maxEg : std::max(0, maxEg - BA_THREADS);
            P.slots = all_lds ? 0 : (maxEg > BA_THREADS ? 2 : 1);
            for (int q = 1; q <= max_seq && !fits; ++q) {
                nseq = q;
                // (pieces side by side in every chunk: the chains of a chunk are half as long, the idle waves take the other half)
                npar = do_schur ? std::max(1, BA_WAVES / npair) : 1;
                if (q > 1 && g_ba_chunk_pieces == 1) npar = 1;
                if (env_nsplit && q == 1) npar = std::min(npar, env_nsplit);
                nsplit = nseq * npar;
                const int msplit = (msteps + nsplit - 1) / nsplit;
                // passes of the landmark-block phase: as few as keep its staged rows inside what the U chunk needs anyway
                const size_t floor_area = ba_uarea_doubles(do_schur ? 4 * npar * msplit : 0, ldu, 0, maxEpose, nhp, G, p->fix_points);
                int npt = 1, pt_edges = maxEg;
                for (; npt <= 8; ++npt) {
                    pt_edges = 0;
                    for (int g = 0; g < G; ++g) {
                        const int Lg_ = wg_pt[g + 1] - wg_pt[g];
                        for (int h = 0; h < npt; ++h) {
                            int cnt = 0;
                            for (int ll = wg_pt[g] + (int)((long long)Lg_ * h / npt); ll < wg_pt[g] + (int)((long long)Lg_ * (h + 1) / npt); ++ll) cnt += deg[ll];
                            pt_edges = std::max(pt_edges, cnt);
                        }
                    }
                    if ((size_t)BA_SXS * pt_edges <= floor_area || npt == 8) break;
                }
                P.npt = p->fix_points ? 1 : npt;
                P.panel = (n + 1 > 32 || g_ba_block_solver) ? 1 : 0;
                uarea = ba_uarea_doubles(do_schur ? 4 * npar * msplit : 0, ldu, pt_edges, maxEpose, nhp, G, p->fix_points);
                // ranges of more than 512 edges (the throughput cut) MAY keep their measurements in device memory (read once or twice
                // per trial from L2): 16 bytes per edge of LDS then go to the U chunks instead (BA5 on 13 workgroups: two chunks where
                // three were needed).  Tried only when the LDS form does not fit this number of chunks (BA5 on 14 workgroups has the
                // room for both: window 2.76 vs 2.79 ms with the measurements in LDS).
                const bool uv_may_be_global = !all_lds && maxEg > BA_THREADS && g_ba_uv_global;
                for (int uvg = 0; uvg <= (uv_may_be_global ? 1 : 0) && !fits; ++uvg) {
                    P.uv_global = uvg;
                    // one column piece per chunk (no split tiles while U is live): the reduced system can live in the U area
                    P.alias_sl = (do_schur && npar == 1 && g_ba_alias_sl && uarea >= ba_solver_matrix_doubles(n, nlow + nhp, G, npair, npar)) ? 1 : 0;
                    P.lds = ba_lds_bytes(F, n, nlow, nhp, G, npair, npar, nfree, maxEg, maxLg, p->fix_points, uarea, P.e2_edges, P.panel, P.uv_global,
                                         P.alias_sl);
                    fits = P.lds <= BA_LDS_BUDGET;
                }
                if (plan_trace)
                    std::fprintf(stderr, "[mvo plan] G %d rows_in_lds %d chunks %d pieces %d pt_passes %d maxEg %d maxLg %d uarea %zu B lds %zu B (budget %d) %s\n",
                                 G, all_lds, nseq, npar, P.npt, maxEg, maxLg, uarea * 8, P.lds, BA_LDS_BUDGET, fits ? "fits" : "-");
                // rows in LDS only while ONE chunk of U and ONE pass of the pose-block rows still fit next to them: with more
                // chunks / passes the barriers cost more than the registers save (measured: 2.76 vs 2.41 ms on the BA5 window)
                if (all_lds && g_ba_edge_rows != 0) {
                    if (uarea < (size_t)BA_MSTRIDE * (maxEg + 2 * F)) fits = false;
                    break;
                }
            }
        }
        if (fits && maxEg <= BA_EDGE_SLOTS * BA_THREADS && maxLg < 32000) break;
        if (G >= BA_MAX_WGS)
            return mvo_set_err(ctx, MVO_ERR_CAPACITY, "BA window too large for the LDS-resident solver", hipSuccess);
        {   // next larger candidate: multiples of the per-XCD allowance and powers of two
            int next = BA_MAX_WGS;
            for (int c : {2, 4, 8, 16, 32, 64, 128, 256, per_xcd / 2, per_xcd, 2 * per_xcd, 4 * per_xcd, 8 * per_xcd})
                if (c > G && c < next) next = c;
            G = next;
        }
    }
    g_stage_times.lap(0);
    P.nsplit = nsplit;
    P.npar = npar;
    P.nseq = nseq;
    P.uarea = uarea;
    P.G = G;
    P.maxEg = maxEg;
    P.maxLg = maxLg;
    // A window of more than one XCD's worth of workgroups sums its Schur partials per XCD first (group k = the workgroups
    // g = k mod K, placed on one XCD by the launch: stride 8 / K), then the K group sums: the bulk of the exchange stays in
    // an L2.  The grouping is part of the summation plan (ba_get_plan), never a matter of where the workgroups ended up.
    P.groups = 1;
    if (G > 32 && g_ba_groups) {
        int K = 2;
        while (G / K > 32 && K < 8) K *= 2;
        if (G % K == 0 && G / K <= 32 && uarea >= (size_t)K * (size_t)(nlow + nhp)) P.groups = K;
    }
    P.slice = (nlow + G / P.groups - 1) / (G / P.groups);
    // ---- edges sorted by (owner workgroup, pose); adjacency tables
    std::vector<int>&e_pose = SC.e_pose, &e_point = SC.e_point, &ptstart = SC.ptstart, &ptlist = SC.ptlist;
    std::vector<double>& e_uv = SC.e_uv;
    e_pose.resize(E);
    e_point.resize(E);
    ptstart.assign(L + 1, 0);
    ptlist.resize(E);
    e_uv.resize(2 * (size_t)E);
    {
        std::vector<int>& cur = SC.cur;
        cur.resize((size_t)G * std::max(F, 1));
        for (int g = 0; g < G; ++g)
            for (int q = 0; q < F; ++q) cur[(size_t)g * F + q] = wg_pose[(size_t)g * (F + 1) + q];
        for (int e : act) {
            const int k = cur[(size_t)owner[p->edge_point[e]] * F + p->edge_pose[e]]++;
            e_pose[k] = p->edge_pose[e];
            e_point[k] = p->edge_point[e];
            e_uv[2 * (size_t)k] = p->edge_uv[2 * (size_t)e];
            e_uv[2 * (size_t)k + 1] = p->edge_uv[2 * (size_t)e + 1];
        }
    }
    for (int k = 0; k < E; ++k) ptstart[e_point[k] + 1]++;
    for (int i = 0; i < L; ++i) ptstart[i + 1] += ptstart[i];
    {
        std::vector<int>& cur = SC.cur2;
        cur.assign(ptstart.begin(), ptstart.end() - 1);
        for (int k = 0; k < E; ++k) ptlist[cur[e_point[k]]++] = k;
    }
    // first observation of every (landmark, pose slot) pair and the rank of every edge among the observations of its pair
    // (ascending edge order: the order in which the oracle adds them)
    std::vector<short>&eof = SC.eof, &dup = SC.dup;
    eof.assign((size_t)std::max(L, 1) * std::max(nfree, 1), -1);
    dup.assign(std::max(E, 1), 0);
    int max_dup = 0;
    {
        std::vector<short>& seen = SC.seen;
        seen.assign((size_t)std::max(L, 1) * std::max(nfree, 1), 0);
        for (int k = 0; k < E; ++k) {
            const int sl = pose_slot[e_pose[k]];
            if (sl < 0) continue;
            const size_t q = (size_t)e_point[k] * nfree + sl;
            if (eof[q] < 0) eof[q] = (short)(k - wg_edge[owner[e_point[k]]]);
            dup[k] = seen[q]++;
            max_dup = std::max(max_dup, (int)dup[k]);
        }
    }
    P.max_dup = max_dup;
    // ---- packed order of the reduced system and its place inside the 16 x 16 tile pairs
    std::vector<short>& pkt = SC.pkt;  // (depends on n only: kept from the last window of this thread)
    if (SC.pkt_n != n) {
        SC.pkt_n = n;
        pkt.assign((size_t)npair * 256, -1);
        int pr = 0;
        for (int ti = 0; ti < NT; ++ti)
            for (int tj = ti; tj < NT; ++tj, ++pr)
                for (int r = 0; r < 16; ++r)
                    for (int cidx = 0; cidx < 16; ++cidx) {
                        const int j = 16 * ti + r, i = 16 * tj + cidx;  // entry (smaller, larger) index
                        if (j > i || i > n || (i == n && j >= n)) continue;
                        pkt[(size_t)pr * 256 + r * 16 + cidx] = (short)(i < n ? i * (i + 1) / 2 + j : n * (n + 1) / 2 + j);
                    }
    }

    // ---- layout
    // The exchange areas come FIRST: their sizes depend on (G, n) only, so windows of one shape find them where the last one
    // left them -- memory that only ever held granules, matched by tag.  When their layout does change they are cleared
    // (below): bytes that once held ordinary data must never be taken for a published value.
    Carver cv;
    P.o_xp = cv.take((size_t)G * npk * 16);
    P.o_xr = cv.take((size_t)npk * 16 * (P.groups > 1 ? 2 * P.groups : 1));
    P.o_xh = cv.take((size_t)G * nhp * 16);
    P.o_xc = cv.take((size_t)2 * G * 4 * 8);
    P.x_end = cv.off;
    P.o_desc = cv.take(sizeof(BaDev));
    P.o_pin = cv.take((size_t)F * 128);
    P.o_ptsin = cv.take((size_t)L * 24);
    P.o_wpt = cv.take((size_t)(G + 1) * 4);
    P.o_wed = cv.take((size_t)(G + 1) * 4);
    P.o_wps = cv.take((size_t)G * (F + 1) * 4);
    P.o_ep = cv.take((size_t)E * 4 + 4);
    P.o_el = cv.take((size_t)E * 4 + 4);
    P.o_uv = cv.take((size_t)E * 16 + 16);
    P.o_ptstart = cv.take((size_t)(L + 1) * 4);
    P.o_ptl = cv.take((size_t)E * 4 + 4);
    P.o_eof = cv.take(eof.size() * 2);
    P.o_dup = cv.take(dup.size() * 2);
    P.o_slot = cv.take((size_t)F * 4 + 4);
    P.o_sp = cv.take((size_t)nfree * 4 + 4);
    P.o_pkt = cv.take(pkt.size() * 2);
    P.upload_end = cv.off;
    P.o_stats = cv.take(sizeof(BaStatsDev));
    P.o_pout = cv.take((size_t)F * 128);
    P.o_pts = cv.take((size_t)L * 24);
    P.o_uvd = cv.take((size_t)E * 16 + 16);
    P.o_dxl = cv.take((size_t)L * 24 + 16);
    P.total = cv.off;
    Carver pc;
    pc.off = P.upload_end;
    P.m_stats = pc.take(sizeof(BaStatsDev));
    P.m_poses = pc.take((size_t)std::max(F, 1) * 128);
    P.m_pts = pc.take((size_t)std::max(L, 1) * 24);
    P.m_trace = pc.take(sizeof(BaTraceRow) * BA_TRACE_MAX);
    P.pin_total = pc.off;
    g_stage_times.lap(1);
    int r = ws_reserve(ctx, ws, P.total, P.pin_total);
    if (r) return r;
    const size_t sig[5] = {P.o_xp, P.o_xr, P.o_xh, P.o_xc, P.x_end};
    const bool moved = std::memcmp(sig, ws.x_sig, sizeof sig) != 0;
    if (++ws.seq >= (1u << 20) || moved) {  // tag space exhausted / exchange areas laid out anew: clean exchange memory
        MVO_HIP(hipMemsetAsync(ws.dev, 0, std::max(P.x_end, ws.x_sig[4]), ctx->stream));
        // (a window for the resident grid is not ordered behind this stream: wait here; layout changes are rare)
        if (svc) MVO_HIP(hipStreamSynchronize(ctx->stream));
        if (ws.seq >= (1u << 20)) ws.seq = 1;
        std::memcpy(ws.x_sig, sig, sizeof sig);
    }

    g_stage_times.lap(2);
    // ---- upload image
    char* h = ws.pin;
    char* D = ws.dev;
    if (F) std::memcpy(h + P.o_pin, p->pose_T_w_c, (size_t)F * 128);
    if (L) std::memcpy(h + P.o_ptsin, p->points, (size_t)L * 24);
    std::memcpy(h + P.o_wpt, wg_pt.data(), (size_t)(G + 1) * 4);
    std::memcpy(h + P.o_wed, wg_edge.data(), (size_t)(G + 1) * 4);
    std::memcpy(h + P.o_wps, wg_pose.data(), wg_pose.size() * 4);
    if (E) {
        std::memcpy(h + P.o_ep, e_pose.data(), (size_t)E * 4);
        std::memcpy(h + P.o_el, e_point.data(), (size_t)E * 4);
        std::memcpy(h + P.o_uv, e_uv.data(), (size_t)E * 16);
        std::memcpy(h + P.o_ptl, ptlist.data(), (size_t)E * 4);
    }
    std::memcpy(h + P.o_ptstart, ptstart.data(), (size_t)(L + 1) * 4);
    std::memcpy(h + P.o_eof, eof.data(), eof.size() * 2);
    std::memcpy(h + P.o_dup, dup.data(), dup.size() * 2);
    if (F) std::memcpy(h + P.o_slot, pose_slot.data(), (size_t)F * 4);
    if (nfree) std::memcpy(h + P.o_sp, slot_pose.data(), (size_t)nfree * 4);
    std::memcpy(h + P.o_pkt, pkt.data(), pkt.size() * 2);
    BaDev B{};
    B.F = F;
    B.L = L;
    B.E = E;
    B.G = G;
    B.nfree = nfree;
    B.n = n;
    B.NT = NT;
    B.npair = npair;
    B.fix_points = P.fix_points;
    B.max_it = p->max_iterations;
    B.maxEg = maxEg;
    B.maxLg = maxLg;
    B.max_dup = max_dup;
    B.nlow = nlow;
    B.npk = npk;
    B.slice = P.slice;
    B.nsplit = nsplit;
    B.npar = npar;
    B.nseq = nseq;
    B.uarea = (int)uarea;
    B.e2_edges = P.e2_edges;
    B.uv_global = P.uv_global;
    B.alias_sl = P.alias_sl;
    B.groups = P.groups;
    B.uv_dev = (double*)(D + P.o_uvd);
    B.dxl_dev = (double*)(D + P.o_dxl);
    B.npt = P.npt;
    B.panel = P.panel;
    B.slots = P.slots;
    B.ldu = ldu;
    B.nhp = nhp;
    B.f = p->focal;
    B.cx = p->cx;
    B.cy = p->cy;
    B.delta = p->huber_delta;
    B.lc00 = lc00;
    B.lc01 = lc01;
    B.lc11 = lc11;
    // Windows for the resident solver service are read by the device straight from this pinned image (the grid is already
    // running: a host-to-device copy into memory it may have cached would not be seen); every other window is uploaded.
    // (pose-only windows solve in a fraction of a millisecond: holding CUs resident for them would only take them from
    // the callers' other kernels -- they keep the launch path)
    P.service = svc && P.slots != 0 && G <= service_for(ctx->device).wgs_per_slot.load(std::memory_order_relaxed);
    char* I = P.service ? ws.pin : D;  // where the kernel finds the inputs
    B.poses_in = (const double*)(I + P.o_pin);
    B.poses_out = (double*)(D + P.o_pout);
    B.pts_in = (const double*)(I + P.o_ptsin);
    B.pts_out = (double*)(D + P.o_pts);
    B.wg_pt_start = (const int*)(I + P.o_wpt);
    B.wg_edge_start = (const int*)(I + P.o_wed);
    B.wg_pose_start = (const int*)(I + P.o_wps);
    B.e_pose = (const int*)(I + P.o_ep);
    B.e_point = (const int*)(I + P.o_el);
    B.e_uv = (const double*)(I + P.o_uv);
    B.pt_edge_start = (const int*)(I + P.o_ptstart);
    B.pt_edge_list = (const int*)(I + P.o_ptl);
    B.eof = (const short*)(I + P.o_eof);
    B.dup_rank = (const short*)(I + P.o_dup);
    B.pose_slot = (const int*)(I + P.o_slot);
    B.slot_pose = (const int*)(I + P.o_sp);
    B.pk_of_tile = (const short*)(I + P.o_pkt);
    B.xP = (ba_u64*)(D + P.o_xp);
    B.xR = (ba_u64*)(D + P.o_xr);
    B.xH = (ba_u64*)(D + P.o_xh);
    B.xC = (ba_u64*)(D + P.o_xc);
    B.stats = (BaStatsDev*)(D + P.o_stats);
    B.h_stats = (BaStatsDev*)(ws.pin + P.m_stats);
    B.h_poses = (double*)(ws.pin + P.m_poses);
    B.h_pts = (L && !p->fix_points) ? (double*)(ws.pin + P.m_pts) : nullptr;
    B.trace = want_trace ? (BaTraceRow*)(ws.pin + P.m_trace) : nullptr;
    std::memcpy(h + P.o_desc, &B, sizeof(B));
    std::memset(ws.pin + P.m_stats, 0, sizeof(BaStatsDev));
    g_stage_times.lap(3);
    g_stage_times.done();
    return MVO_OK;
}

int ba_upload(mvo_ctx* ctx, BaWorkspace& ws) {
    if (ws.plan.service) return MVO_OK;  // (the resident grid reads the pinned image itself)
    MVO_HIP(hipMemcpyAsync(ws.dev + ws.plan.x_end, ws.pin + ws.plan.x_end, ws.plan.upload_end - ws.plan.x_end, hipMemcpyHostToDevice, ctx->stream));
    MVO_HIP(hipEventRecord(ws.ready, ctx->stream));
    return MVO_OK;
}

// after the launch completed: pinned mirrors -> caller buffers
int ba_collect(mvo_ctx* ctx, BaWorkspace& ws, const BaJob& job, double* poses, double* points, mvo_ba_stats* st) {
    const BaPlan& P = ws.plan;
    if (st) std::memset(st, 0, sizeof(*st));
    if (!P.runnable) return MVO_OK;
    if (job.err != hipSuccess) return mvo_set_err(ctx, MVO_ERR_HIP, "k_ba_lm launch", job.err);
    const BaStatsDev* s = (const BaStatsDev*)(ws.pin + P.m_stats);
    for (int i = 0; i < BA_NPHASE && i < 16; ++i) ctx->ba_phase[i] = s->phase[i];
    ctx->ba_wgs = P.G;
    ctx->ba_trials = s->trials;
    if (ctx->prof) {
        ProfEntry& pe = ctx->prof_acc["k_ba_lm"];
        pe.launches += 1;
        pe.ms += job.ms;
    }
    if (s->error) return mvo_set_err(ctx, MVO_ERR_HIP, "BA hand-off timed out (workgroups not co-resident)", hipSuccess);
    if (poses && P.F) std::memcpy(poses, ws.pin + P.m_poses, (size_t)P.F * 128);
    if (points && P.L && !P.fix_points) std::memcpy(points, ws.pin + P.m_pts, (size_t)P.L * 24);
    if (st) {
        st->iterations = s->iterations;
        st->trials = s->trials;
        st->terminated = s->terminated;
        st->failed_solves = s->failed_solves;
        st->stale_steps = s->stale_steps;
        st->chi2_initial = s->chi2_initial;
        st->chi2_final = s->chi2_final;
        st->lambda_final = s->lambda_final;
    }
    return MVO_OK;
}

}  // namespace

// ctx-owned pool of workspaces (index 0: mvo_bundle_adjustment; more for mvo_ba_solve_batch) + the pending job of the
// begin / end pair
struct mvo_ba_pool {
    std::vector<BaWorkspace*> ws;
    BaJob pending;
    bool has_pending = false;
    int want_trace = 0;
};
static mvo_ba_pool* pool_of(mvo_ctx* ctx) {
    if (!ctx->ba_pool) ctx->ba_pool = new mvo_ba_pool();
    return ctx->ba_pool;
}
static BaWorkspace* pool_ws(mvo_ctx* ctx, size_t i) {
    mvo_ba_pool* p = pool_of(ctx);
    while (p->ws.size() <= i) p->ws.push_back(new BaWorkspace());
    return p->ws[i];
}
void ba_pool_release(mvo_ctx* ctx) {
    if (!ctx->ba_pool) return;
    // a ctx destroyed between mvo_bundle_adjustment_begin and _end still has a window queued or running in the launch
    // service, which reads the workspace and writes its pinned mirrors: wait for it before anything is freed
    mvo_ba_pool* pool = ctx->ba_pool;
    if (pool->has_pending && pool->pending.ws && pool->pending.ws->plan.runnable && !pool->pending.done)
        service_wait(service_for(ctx->device), &pool->pending, 1);
    pool->has_pending = false;
    for (BaWorkspace* w : ctx->ba_pool->ws) {
        ws_free(*w);
        delete w;
    }
    delete ctx->ba_pool;
    ctx->ba_pool = nullptr;
}
void ba_set_trace(mvo_ctx* ctx, int on) { pool_of(ctx)->want_trace = on; }

// optimization::bundleAdjustment, first half: graph -> window -> upload -> launch (asynchronous)
int ba_begin_device(mvo_ctx* ctx, const mvo_ba_problem* p) {
    mvo_ba_pool* pool = pool_of(ctx);
    if (pool->has_pending) return mvo_set_err(ctx, MVO_ERR_STATE, "a bundle adjustment is already in flight on this ctx", hipSuccess);
    BaWorkspace* ws = pool_ws(ctx, 0);
    int r = ba_stage(ctx, p, *ws, pool->want_trace != 0);
    if (r) return r;
    pool->pending = BaJob();
    pool->pending.ws = ws;
    pool->pending.use_mfma = g_ba_use_mfma;
    pool->has_pending = true;
    if (!ws->plan.runnable) {
        pool->pending.done = true;
        return MVO_OK;
    }
    if ((r = ba_upload(ctx, *ws))) {
        pool->has_pending = false;
        return r;
    }
    service_submit(service_for(ctx->device), &pool->pending, 1);
    return MVO_OK;
}
// second half: wait, read back (g2o_ba.cpp:298-316)
int ba_end_device(mvo_ctx* ctx, mvo_ba_problem* p, mvo_ba_stats* st) {
    mvo_ba_pool* pool = pool_of(ctx);
    if (!pool->has_pending) return mvo_set_err(ctx, MVO_ERR_STATE, "no bundle adjustment in flight on this ctx", hipSuccess);
    BaWorkspace* ws = pool->pending.ws;
    if (ws->plan.runnable) service_wait(service_for(ctx->device), &pool->pending, 1);
    pool->has_pending = false;
    return ba_collect(ctx, *ws, pool->pending, p ? p->pose_T_w_c : nullptr, p ? p->points : nullptr, st);
}
int ba_solve_device(mvo_ctx* ctx, mvo_ba_problem* p, mvo_ba_stats* st) {
    int r = ba_begin_device(ctx, p);
    if (r) return r;
    return ba_end_device(ctx, p, st);
}

// n independent windows in ONE launch (up to BA_MAX_BATCH per grid; more are split over consecutive launches)
int ba_solve_batch_device(mvo_ctx* ctx, mvo_ba_problem* ps, int n, mvo_ba_stats* sts) {
    mvo_ba_pool* pool = pool_of(ctx);
    if (pool->has_pending) return mvo_set_err(ctx, MVO_ERR_STATE, "a bundle adjustment is already in flight on this ctx", hipSuccess);
    std::vector<BaJob> jobs(n);
    std::vector<BaJob*> run;
    for (int i = 0; i < n; ++i) {
        BaWorkspace* ws = pool_ws(ctx, (size_t)i);
        int r = ba_stage(ctx, &ps[i], *ws, false);
        if (r) return r;
        jobs[i].ws = ws;
        jobs[i].use_mfma = g_ba_use_mfma;
        if (!ws->plan.runnable) {
            jobs[i].done = true;
            continue;
        }
        if ((r = ba_upload(ctx, *ws))) return r;
    }
    BaService& S = service_for(ctx->device);
    {
        std::lock_guard<std::mutex> lk(S.m);
        for (int i = 0; i < n; ++i)
            if (!jobs[i].done) {
                S.q.push_back(&jobs[i]);
                S.q_pending.fetch_add(1, std::memory_order_release);
            }
    }
    S.cv_work.notify_one();
    service_wait(S, jobs.data(), n);
    int first = MVO_OK;
    for (int i = 0; i < n; ++i) {
        int r = ba_collect(ctx, *jobs[i].ws, jobs[i], ps[i].pose_T_w_c, ps[i].points, sts ? &sts[i] : nullptr);
        if (r && !first) first = r;
    }
    return first;
}

// ---- resident windows: upload once (mvo_ba_prepare), solve any number of times from the same initial state
struct mvo_ba_handle {
    BaWorkspace ws;
    BaJob job;
    bool launched = false;
    bool uploaded = false;
};
int ba_prepare_device(mvo_ctx* ctx, const mvo_ba_problem* p, mvo_ba_handle** out) {
    *out = nullptr;
    mvo_ba_handle* H = new mvo_ba_handle();
    int r = ba_stage(ctx, p, H->ws, pool_of(ctx)->want_trace != 0);
    if (!r && H->ws.plan.runnable) r = ba_upload(ctx, H->ws);
    if (!r) {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) r = mvo_set_err(ctx, MVO_ERR_HIP, "BA upload", e);
    }
    if (r) {
        ws_free(H->ws);
        delete H;
        return r;
    }
    H->uploaded = true;
    *out = H;
    return MVO_OK;
}
int ba_run_device(mvo_ctx* ctx, mvo_ba_handle* H) {
    if (!H->ws.plan.runnable) return MVO_OK;
    if (H->launched) return mvo_set_err(ctx, MVO_ERR_STATE, "mvo_ba_fetch must follow every mvo_ba_solve_resident", hipSuccess);
    if (++H->ws.seq >= (1u << 20)) H->ws.seq = 1;  // (one window: at most 2^20 re-solves between clean-ups is plenty)
    H->job = BaJob();
    H->job.ws = &H->ws;
    H->job.use_mfma = g_ba_use_mfma;
    MVO_HIP(hipEventRecord(H->ws.ready, ctx->stream));
    service_submit(service_for(ctx->device), &H->job, 1);
    H->launched = true;
    return MVO_OK;
}
int ba_fetch_device(mvo_ctx* ctx, mvo_ba_handle* H, double* poses, double* points, mvo_ba_stats* st) {
    if (H->ws.plan.runnable) {
        if (!H->launched) return mvo_set_err(ctx, MVO_ERR_STATE, "mvo_ba_fetch without mvo_ba_solve_resident", hipSuccess);
        service_wait(service_for(ctx->device), &H->job, 1);
        H->launched = false;
    }
    return ba_collect(ctx, H->ws, H->job, poses, points, st);
}
void ba_release_device(mvo_ctx* ctx, mvo_ba_handle* H) {
    if (!H) return;
    if (H->launched) service_wait(service_for(ctx ? ctx->device : H->ws.device), &H->job, 1);
    ws_free(H->ws);
    delete H;
}

// ---- debug / measurement hooks
int ba_get_trace(mvo_ctx* ctx, mvo_ba_handle* H, double* rows, int cap, int* n) {
    BaWorkspace* ws = H ? &H->ws : (ctx->ba_pool && !ctx->ba_pool->ws.empty() ? ctx->ba_pool->ws[0] : nullptr);
    if (!ws || !ws->pin) return mvo_set_err(ctx, MVO_ERR_STATE, "no bundle adjustment has run", hipSuccess);
    const BaStatsDev* s = (const BaStatsDev*)(ws->pin + ws->plan.m_stats);
    const int m = cap < 0 ? std::min(-cap, BA_TRACE_MAX) : std::min(std::min(s->trials, BA_TRACE_MAX), cap);  // cap < 0: raw rows
    if (rows) std::memcpy(rows, ws->pin + ws->plan.m_trace, (size_t)m * sizeof(BaTraceRow));
    if (n) *n = m;
    return MVO_OK;
}
int ba_get_plan(mvo_ctx* ctx, mvo_ba_handle* H, int* G, int* nsplit, int32_t* wg_pt, int cap) {
    BaWorkspace* ws = H ? &H->ws : (ctx->ba_pool && !ctx->ba_pool->ws.empty() ? ctx->ba_pool->ws[0] : nullptr);
    if (!ws) return mvo_set_err(ctx, MVO_ERR_STATE, "no bundle adjustment has been planned", hipSuccess);
    if (G) *G = ws->plan.G;
    if (nsplit) *nsplit = ws->plan.nsplit | (ws->plan.groups > 1 ? ws->plan.groups << 16 : 0);  // (bits 16 ..: the groups of the Schur exchange when there are several)
    if (wg_pt)
        for (int i = 0; i <= ws->plan.G && i < cap; ++i) wg_pt[i] = ws->plan.wg_pt[i];
    return MVO_OK;
}
// wall-clock of the launch thread's stages since the last reset: waiting for work, waiting for a full batch, building +
// issuing the launch, waiting for the kernel, publishing the results (ms)
void ba_service_times(int device, double* out5) {
    for (int i = 0; i < 5; ++i) out5[i] = 0;
    if (device < 0 || device >= 16) return;
    BaService* sp;
    {
        std::lock_guard<std::mutex> lk(*g_service_start);
        sp = g_service[device & 15];
    }
    if (!sp) return;
    std::lock_guard<std::mutex> lk(sp->m);
    out5[0] = sp->t_idle, out5[1] = sp->t_batch, out5[2] = sp->t_launch, out5[3] = sp->t_sync, out5[4] = sp->t_post;
}
// mvo_synchronize: nothing of this ctx is in flight any more; if the resident grid is idle it leaves right away, so that a
// device-wide synchronisation issued next by the caller (hipDeviceSynchronize, torch.cuda.synchronize) does not wait for
// its idle timeout
void ba_service_park(int device) {
    if (device < 0 || device >= 16) return;
    BaService* sp;
    {
        std::lock_guard<std::mutex> lk(*g_service_start);
        sp = g_service[device & 15];
    }
    if (!sp) return;
    std::unique_lock<std::mutex> lk(sp->m);
    if (!sp->resident) return;
    sp->park_requested = true;
    sp->cv_work.notify_all();
    // wait (bounded) until the grid is gone or has work again
    sp->cv_done.wait_for(lk, std::chrono::milliseconds(50), [&] { return !sp->resident || sp->slots_busy > 0 || !sp->q.empty(); });
}
void ba_resident_stats(int device, long long* windows, long long* grid_starts, double* cycles, long long* path_switches) {
    if (windows) *windows = 0;
    if (grid_starts) *grid_starts = 0;
    if (cycles) *cycles = 0;
    if (path_switches) *path_switches = 0;
    if (device < 0 || device >= 16) return;
    BaService* sp;
    {
        std::lock_guard<std::mutex> lk(*g_service_start);
        sp = g_service[device & 15];
    }
    if (!sp) return;
    std::lock_guard<std::mutex> lk(sp->m);
    if (windows) *windows = sp->resident_jobs;
    if (grid_starts) *grid_starts = sp->resident_starts;
    if (cycles) *cycles = sp->resident_cycles;
    if (path_switches) *path_switches = sp->path_switches;
}
void ba_launch_stats(int device, long long* launches, long long* windows, double* ms, int reset) {
    if (launches) *launches = 0;
    if (windows) *windows = 0;
    if (ms) *ms = 0;
    if (device < 0 || device >= 16) return;
    BaService* sp;
    {
        std::lock_guard<std::mutex> lk(*g_service_start);
        sp = g_service[device & 15];
    }
    if (!sp) return;
    BaService& s = *sp;
    std::lock_guard<std::mutex> lk(s.m);
    if (launches) *launches = s.launches;
    if (windows) *windows = s.windows;
    if (ms) *ms = s.ms;
    if (reset) {
        s.launches = s.windows = 0;
        s.ms = 0;
        s.t_idle = s.t_batch = s.t_launch = s.t_sync = s.t_post = 0;
        s.resident_jobs = s.resident_starts = 0;
        s.resident_cycles = 0;
        s.path_switches = 0;
    }
}

// test hook: the service's demand estimate replayed over a list of submission times (seconds, ascending)
int ba_demand_replay(const double* times, int n, unsigned char* decisions) {
    BaDemand d;
    for (int i = 0; i < n; ++i) decisions[i] = d.submit(times[i]) ? 1 : 0;
    return (int)d.flips;
}
