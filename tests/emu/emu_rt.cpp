// TEST INFRASTRUCTURE -- CPU wavefront emulator runtime behind tests/emu/include/hip/hip_runtime.h (see the header there).
//
// Execution model. A kernel launch runs at once (streams are host-order queues). Its workgroups are handed to a pool of OS threads
// (EMU_THREADS, default 16; a grid of up to that many workgroups is co-resident, which is what the kernels with inter-workgroup spin
// barriers are promised by the occupancy gate they apply: emulated device = EMU_CUS compute units x 2 workgroups). Inside a
// workgroup every lane is a fiber with its own stack; lanes run one at a time until they reach a wave collective (MFMA, shuffle, DPP,
// permute, readlane, ballot, wave_barrier ...), a workgroup barrier, a spin-wait yield, or the end of the kernel. A collective completes
// when every lane of the wave that has not exited has arrived; if no lane of the workgroup can run and a wave is only partly there
// (a collective in divergent control flow), the lanes that did arrive complete it among themselves, as the hardware's EXEC mask would.
#include <hip/hip_runtime.h>

#include <cxxabi.h>
#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <time.h>

#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// EMU_ASAN=1 build (tests/emu/build_emu.py): AddressSanitizer has to be told about every stack switch
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define EMU_ASAN 1
#include <sanitizer/common_interface_defs.h>
#endif
#endif
#ifdef EMU_ASAN
#define ASAN_START(save, bottom, size) __sanitizer_start_switch_fiber((save), (bottom), (size))
#define ASAN_FINISH(save, bottom, size) __sanitizer_finish_switch_fiber((save), (bottom), (size))
#else
#define ASAN_START(save, bottom, size) ((void)0)
#define ASAN_FINISH(save, bottom, size) ((void)0)
#endif

extern "C" void emu_switch(void** save_sp, void* load_sp);
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
.size emu_switch, .-emu_switch
)");

namespace emu {

thread_local const LaneIds* cur_lane_ids = nullptr;
thread_local const BlockIds* cur_block_ids = nullptr;
thread_local void* cur_dyn_smem = nullptr;
thread_local unsigned long long counters[C_N] = {};

namespace {

constexpr size_t STACK_BYTES = 512 * 1024;
enum State { READY, WAIT_WAVE, WAIT_BLOCK, SPIN, DONE };

// one completed collective: what its lanes published, kept until every one of them has entered its NEXT collective (or left the kernel)
struct Result {
    alignas(64) unsigned char tab[64 * XS];
    unsigned long long part = 0;
    int refs = 0;
};
struct Wave {
    alignas(64) unsigned char tab[64 * XS];     // what the waiting lanes have published
    unsigned long long arrived = 0, alive = 0, released = 0;
    const void* site[64] = {};                  // where (return address of the wrapper's call) each waiting lane entered its collective
    Result* res[64] = {};                       // the result a lane was released with
    std::vector<Result*> pool;
    int gen = 0;
    ~Wave() { for (Result* r : pool) delete r; }
};
struct Block;
struct Lane {
    LaneIds ids;
    void* sp = nullptr;
    void* asan_fake = nullptr;
    char* stack = nullptr;
    State state = READY;
    int wait_gen = 0;
    Wave* wave = nullptr;
    Block* blk = nullptr;
};
struct Block {
    BlockIds ids;
    std::vector<Lane> lanes;
    std::vector<Wave> waves;
    int bar_gen = 0, bar_count = 0, alive = 0;
    void* sched_sp = nullptr;
    void* asan_fake = nullptr;
    const void* sched_stack = nullptr;
    size_t sched_stack_size = 0;
    KernelCall call;
    const char* name;
};

thread_local Lane* cur = nullptr;
thread_local Block* cur_blk = nullptr;
thread_local std::vector<char*>* stacks = nullptr;

int env_int(const char* name, int dflt);

char* stack_of(size_t i) {
    if (!stacks) stacks = new std::vector<char*>();
    while (stacks->size() <= i) {
        void* p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) { perror("emu: mmap of a lane stack"); abort(); }
        stacks->push_back(static_cast<char*>(p));
    }
    return (*stacks)[i];
}

void to_sched() {
    Lane* me = cur;
    Block* b = cur_blk;
    ASAN_START(me->state == DONE ? nullptr : &me->asan_fake, b->sched_stack, b->sched_stack_size);
    emu_switch(&me->sp, b->sched_sp);
    ASAN_FINISH(me->asan_fake, nullptr, nullptr);
}
Result* result_get(Wave* w) {
    for (Result* r : w->pool)
        if (r->refs == 0) return r;
    w->pool.push_back(new Result());
    return w->pool.back();
}
void release_group(Wave* w, unsigned long long group) {
    Result* r = result_get(w);
    r->part = group;
    r->refs = __builtin_popcountll(group);
    for (unsigned long long m = group; m; m &= m - 1) {
        const int l = __builtin_ctzll(m);
        memcpy(r->tab + XS * l, w->tab + XS * l, XS);
        w->res[l] = r;
    }
    w->released |= group;
    w->arrived &= ~group;
    w->gen++;
}
unsigned long long group_at(const Wave* w, const void* site) {
    unsigned long long g = 0;
    for (unsigned long long m = w->arrived; m; m &= m - 1) {
        const int l = __builtin_ctzll(m);
        if (w->site[l] == site) g |= 1ull << l;
    }
    return g;
}
// Called when every lane of the wave that is still in the kernel waits in a collective, or when nothing in the workgroup can run and
// part of a wave does (the rest waits at the workgroup barrier or spins). Lanes at DIFFERENT call sites are in different instructions
// (an if / else with a shuffle in each arm, lanes that left a loop early): the hardware issues them one after the other under disjoint
// EXEC masks, in its structured order -- then-arm, else-arm, join; a loop body before the loop's exit --, which the code address
// approximates: the group at the lowest call site completes among its own lanes; the others keep waiting for the lanes it sets free to
// join them (reconvergence). One call site = the whole wave in the common case.
bool complete_earliest(Wave* w) {
    if (!w->arrived) return false;
    const void* first = nullptr;
    for (unsigned long long m = w->arrived; m; m &= m - 1) {
        const void* st = w->site[__builtin_ctzll(m)];
        if (!first || st < first) first = st;
    }
    release_group(w, group_at(w, first));
    return true;
}

[[noreturn]] void lane_exit() {
    Lane* me = cur;
    Block* b = cur_blk;
    Wave* w = me->wave;
    me->state = DONE;
    if (w->res[me->ids.lane]) { w->res[me->ids.lane]->refs--; w->res[me->ids.lane] = nullptr; }
    w->alive &= ~(1ull << me->ids.lane);
    if (w->arrived && w->arrived == w->alive) complete_earliest(w);
    b->alive--;
    if (b->bar_count > 0 && b->bar_count == b->alive) { b->bar_count = 0; b->bar_gen++; }
    to_sched();
    abort();
}
extern "C" void emu_lane_entry() {
    ASAN_FINISH(nullptr, &cur_blk->sched_stack, &cur_blk->sched_stack_size);       // (first time on this stack: learn where the scheduler's is)
    cur_blk->call.run(cur_blk->call.ctx);
    lane_exit();
}

void run_block(Block& b) {
    cur_blk = &b;
    cur_block_ids = &b.ids;
    const size_t n = b.lanes.size();
    for (size_t i = 0; i < n; ++i) {
        char* top = stack_of(i) + STACK_BYTES;
        void** sp = reinterpret_cast<void**>(top);
        *--sp = nullptr;                                        // (keeps the entry's frame 16-byte aligned as after a call)
        *--sp = reinterpret_cast<void*>(&emu_lane_entry);
        for (int r = 0; r < 6; ++r) *--sp = nullptr;
        b.lanes[i].sp = sp;
        b.lanes[i].stack = stack_of(i);
    }
    // EMU_ORDER: the order in which the scheduler visits the lanes (a kernel that is correct must not depend on it: a missing barrier
    // between waves can pass by luck of one order) -- 0 ascending (default), 1 descending, 2 waves descending / lanes ascending,
    // >= 3 a shuffle seeded with the value and the workgroup's id
    static const int order_mode = env_int("EMU_ORDER", 0);
    std::vector<unsigned> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = (unsigned)i;
    if (order_mode == 1) for (size_t i = 0; i < n; ++i) order[i] = (unsigned)(n - 1 - i);
    else if (order_mode == 2) {
        const size_t nw = (n + 63) / 64;
        size_t k = 0;
        for (size_t w = nw; w-- > 0;)
            for (size_t l = 64 * w; l < n && l < 64 * (w + 1); ++l) order[k++] = (unsigned)l;
    } else if (order_mode >= 3) {
        unsigned long long st = 0x9E3779B97F4A7C15ull * (unsigned long long)(order_mode + 1) + b.ids.bid.x * 1315423911ull + b.ids.bid.y * 2654435761ull;
        for (size_t i = n; i > 1; --i) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            std::swap(order[i - 1], order[st % i]);
        }
    }
    long idle_spins = 0;
    time_t spin_t0 = 0;
    while (b.alive > 0) {
        bool progressed = false, spinning = false;
        for (size_t oi = 0; oi < n; ++oi) {
            Lane& l = b.lanes[order[oi]];
            bool run = false;
            switch (l.state) {
                case READY: run = true; break;
                case WAIT_WAVE: run = (l.wave->released >> l.ids.lane) & 1ull; break;
                case WAIT_BLOCK: run = b.bar_gen != l.wait_gen; break;
                case SPIN: run = true; spinning = true; break;
                case DONE: break;
            }
            if (!run) continue;
            if (l.state != SPIN) progressed = true;
            l.state = READY;
            cur = &l;
            cur_lane_ids = &l.ids;
            ASAN_START(&b.asan_fake, l.stack, STACK_BYTES);
            emu_switch(&b.sched_sp, l.sp);
            ASAN_FINISH(b.asan_fake, nullptr, nullptr);
        }
        if (progressed) { idle_spins = 0; continue; }
        if (b.alive == 0) break;
        if (spinning) {
            // a lane waits for ANOTHER workgroup (an OS thread that may be descheduled for a while on a loaded host): that is not a deadlock
            // of this workgroup, and nothing here may be released on its account. Only a wall-clock limit ends it (EMU_SPIN_TIMEOUT_S, 900).
            static const int limit_s = env_int("EMU_SPIN_TIMEOUT_S", 900);
            if (idle_spins++ == 0) spin_t0 = time(nullptr);
            if ((idle_spins & 0xFFF) == 0 && time(nullptr) - spin_t0 > limit_s) {
                fprintf(stderr, "emu: kernel %s, workgroup %u: spin-wait on another workgroup for more than %d s\n", b.name, b.ids.bid.x, limit_s);
                abort();
            }
            if (idle_spins > 64) { timespec ts{0, 50000}; nanosleep(&ts, nullptr); } else sched_yield();
            continue;
        }
        // nobody can run: a collective reached by part of a wave completes among the lanes that are there
        bool released = false;
        for (Wave& w : b.waves)
            if (complete_earliest(&w)) released = true;
        if (released) continue;
        if (b.bar_count > 0) { b.bar_count = 0; b.bar_gen++; continue; }
        fprintf(stderr, "emu: deadlock in kernel %s, workgroup %u (%d lanes alive)\n", b.name, b.ids.bid.x, b.alive);
        abort();
    }
    cur = nullptr;
    cur_lane_ids = nullptr;
}

// ---- worker pool ----
struct Job {
    dim3 grid, block;
    size_t shmem;
    const char* name;
    KernelCall call;
    std::atomic<long> next{0};
    long total = 0;
    std::atomic<unsigned long long> counts[C_N];
};
struct KernelCounts { unsigned long long launches = 0, workgroups = 0, c[C_N] = {}; };
std::map<std::string, KernelCounts>& kernel_counts = *new std::map<std::string, KernelCounts>();
std::mutex& counts_mu = *new std::mutex();
// (never destroyed: the detached workers wait on them until the process ends)
std::mutex& mu = *new std::mutex();
std::condition_variable& cv_work = *new std::condition_variable();
std::condition_variable& cv_done = *new std::condition_variable();
Job* job = nullptr;
long job_seq = 0;
int workers_busy = 0;
std::vector<std::thread>* pool = nullptr;

void work_on(Job& j) {
    const unsigned nthreads = j.block.x * j.block.y * j.block.z;
    const unsigned nwaves = (nthreads + 63) / 64;
    char* dyn = nullptr;
    if (j.shmem) {
        if (posix_memalign(reinterpret_cast<void**>(&dyn), 64, j.shmem)) abort();
    }
    for (;;) {
        const long idx = j.next.fetch_add(1);
        if (idx >= j.total) break;
        Block b;
        b.ids.gdim = Idx{j.grid.x, j.grid.y, j.grid.z};
        b.ids.bdim = Idx{j.block.x, j.block.y, j.block.z};
        b.ids.bid = Idx{(unsigned)(idx % j.grid.x), (unsigned)(idx / j.grid.x % j.grid.y), (unsigned)(idx / ((long)j.grid.x * j.grid.y))};
        b.call = j.call;
        b.name = j.name;
        b.lanes.resize(nthreads);
        b.waves.resize(nwaves);
        b.alive = (int)nthreads;
        for (unsigned t = 0; t < nthreads; ++t) {
            Lane& l = b.lanes[t];
            l.ids.tid = Idx{t % j.block.x, t / j.block.x % j.block.y, t / (j.block.x * j.block.y)};
            l.ids.lane = (int)(t & 63);
            l.wave = &b.waves[t >> 6];
            l.blk = &b;
            l.wave->alive |= 1ull << (t & 63);
        }
        if (dyn) memset(dyn, 0xFF, j.shmem);          // LDS holds garbage at the start of a workgroup: NaNs / -1 here
        cur_dyn_smem = dyn;
        run_block(b);
    }
    free(dyn);
    for (int i = 0; i < C_N; ++i) { j.counts[i].fetch_add(counters[i]); counters[i] = 0; }
}

std::string kernel_name(const void* fn, const char* text) {
    Dl_info info;
    if (fn && dladdr(fn, &info) && info.dli_sname) {
        int status = 0;
        char* d = abi::__cxa_demangle(info.dli_sname, nullptr, nullptr, &status);
        std::string out = status == 0 && d ? d : info.dli_sname;
        free(d);
        const size_t paren = out.rfind('(');                 // drop the parameter list and the namespace
        if (paren != std::string::npos) out.resize(paren);
        if (out.rfind("void ", 0) == 0) out.erase(0, 5);
        if (out.rfind("refil::", 0) == 0) out.erase(0, 7);
        return out;
    }
    return text;
}

void worker_main() {
    long seen = 0;
    for (;;) {
        Job* j;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_work.wait(lk, [&] { return job_seq != seen; });
            seen = job_seq;
            j = job;
        }
        if (j) work_on(*j);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (--workers_busy == 0) cv_done.notify_all();
        }
    }
}
int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}
int pool_size() {
    static const int n = [] { int v = env_int("EMU_THREADS", 16); return v < 1 ? 1 : v; }();
    return n;
}
std::mutex& launch_mu = *new std::mutex();     // one launch at a time (the host API is callable from several threads)

}  // namespace

__attribute__((noinline)) Xchg xchg(const void* mine, int nbytes) {
    Lane* me = cur;
    Wave* w = me->wave;
    const int lane = me->ids.lane;
    const unsigned long long bit = 1ull << lane;
    if (w->res[lane]) { w->res[lane]->refs--; w->res[lane] = nullptr; }        // (the previous collective's result has been read)
    if (nbytes) memcpy(w->tab + XS * lane, mine, (size_t)nbytes);
    w->site[lane] = __builtin_return_address(0);
    w->arrived |= bit;
    if (w->arrived == w->alive) complete_earliest(w);
    if (!(w->released & bit)) { me->state = WAIT_WAVE; to_sched(); }
    w->released &= ~bit;
    return Xchg{w->res[lane]->tab, w->res[lane]->part};
}
void block_sync() {
    Lane* me = cur;
    Block* b = cur_blk;
    const int mygen = b->bar_gen;
    if (++b->bar_count == b->alive) { b->bar_count = 0; b->bar_gen++; }
    else { me->state = WAIT_BLOCK; me->wait_gen = mygen; to_sched(); }
}
void lane_yield() {
    cur->state = SPIN;
    to_sched();
}
unsigned long long wall_ticks() {
    // 100 MHz on the hardware; the emulator is orders of magnitude slower than the GPU, so its clock runs 1000 x slower too
    // (the kernels' wall-clock time-outs then scale with the emulation)
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ((unsigned long long)ts.tv_sec * 1000000000ull + (unsigned long long)ts.tv_nsec) / 10000ull;
}

void launch(dim3 grid, dim3 block, size_t shmem, const char* name, const void* fn, KernelCall call) {
    std::lock_guard<std::mutex> lg(launch_mu);
    Job j;
    for (int i = 0; i < C_N; ++i) j.counts[i].store(0);
    j.grid = grid; j.block = block; j.shmem = shmem; j.name = name; j.call = call;
    j.total = (long)grid.x * grid.y * grid.z;
    if (j.total <= 0 || block.x * block.y * block.z == 0) return;
    if (env_int("EMU_TRACE", 0)) fprintf(stderr, "emu: launch %s grid %u x %u x %u block %u lds %zu\n", name, grid.x, grid.y, grid.z, block.x * block.y * block.z, shmem);
    const int n = pool_size();
    std::unique_lock<std::mutex> lk(mu);
    if (!pool) {
        pool = new std::vector<std::thread>();
        for (int i = 0; i < n; ++i) { pool->emplace_back(worker_main); pool->back().detach(); }
    }
    job = &j;
    workers_busy = n;
    ++job_seq;
    cv_work.notify_all();
    cv_done.wait(lk, [&] { return workers_busy == 0; });
    job = nullptr;
    lk.unlock();
    static const bool counting = env_int("EMU_COUNT", 0) != 0;
    if (counting) {
        std::lock_guard<std::mutex> cg(counts_mu);
        KernelCounts& kc = kernel_counts[kernel_name(fn, name)];
        kc.launches++;
        kc.workgroups += (unsigned long long)j.total;
        for (int i = 0; i < C_N; ++i) kc.c[i] += j.counts[i].load();
    }
}

}  // namespace emu

// EMU_COUNT=1: per kernel instantiation -- launches, workgroups, wave-level matrix instructions by type, raw-buffer bytes -- as text lines
// "name launches workgroups mfma_32x32x2_f32 mfma_16x16x4_f32 mfma_4x4x1_f32 mfma_16x16x32_bf16 mfma_32x32x16_bf16 buf_load_bytes buf_store_bytes"
extern "C" int emu_counters_dump(char* buf, int n, int reset) {
    std::lock_guard<std::mutex> cg(emu::counts_mu);
    std::string out;
    for (auto& kv : emu::kernel_counts) {
        out += kv.first + "\t" + std::to_string(kv.second.launches) + "\t" + std::to_string(kv.second.workgroups);
        for (int i = 0; i < emu::C_N; ++i) out += "\t" + std::to_string(kv.second.c[i]);
        out += "\n";
    }
    if (reset) emu::kernel_counts.clear();
    if (buf && n > 0) { strncpy(buf, out.c_str(), (size_t)n - 1); buf[n - 1] = 0; }
    return (int)out.size() + 1;
}

// ---------------------------------------------------------------- host API ----
struct emuStream { int id; };
struct emuEvent { double t_ms; };

static double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

const char* hipGetErrorString(hipError_t e) {
    switch (e) {
        case hipSuccess: return "success";
        case hipErrorInvalidValue: return "invalid value";
        case hipErrorOutOfMemory: return "out of memory";
        case hipErrorNotSupported: return "not supported by the CPU emulator";
        default: return "error";
    }
}
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) {
    if (a == hipDeviceAttributeMultiprocessorCount) { *v = emu::env_int("EMU_CUS", 8); return hipSuccess; }
    *v = 0;
    return hipErrorInvalidValue;
}
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { return posix_memalign(p, 256, n ? n : 1) ? hipErrorOutOfMemory : hipSuccess; }
hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyFromSymbol(void* d, const void* sym, size_t n, size_t off, hipMemcpyKind) { memcpy(d, static_cast<const char*>(sym) + off, n); return hipSuccess; }
hipError_t hipMemcpyToSymbol(const void* sym, const void* s, size_t n, size_t off, hipMemcpyKind) { memcpy(const_cast<char*>(static_cast<const char*>(sym)) + off, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < height; ++r) memmove(static_cast<char*>(d) + r * dpitch, static_cast<const char*>(s) + r * spitch, width);
    return hipSuccess;
}
hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static std::atomic<int> g_stream_ids{1};
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = new emuStream{g_stream_ids++}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreateWithPriority(s, 0, 0); }
hipError_t hipStreamCreate(hipStream_t* s) { return hipStreamCreateWithPriority(s, 0, 0); }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { if (g) *g = nullptr; return hipErrorNotSupported; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emuEvent{0.0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { if (e) e->t_ms = now_ms(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t*, void*) { return hipErrorNotSupported; }
hipError_t hipIpcOpenMemHandle(void**, hipIpcMemHandle_t, unsigned) { return hipErrorNotSupported; }
hipError_t hipIpcCloseMemHandle(void*) { return hipErrorNotSupported; }
