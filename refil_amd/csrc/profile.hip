#include "profile.h"

#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "../../include/refil_hip.h"

namespace refil {

constexpr int MAX_SLOTS = 1 << 14;   // clocked launches between enable and collect
constexpr size_t SLOT_U64 = (size_t)CLK_LINES * CLK_LINE_U64;   // 2 KB per launch (common.h: ClkScope)

struct Rec { const char* name; double flops, bytes; hipEvent_t e0, e1; bool launched; int slot; };
static int g_mode = 0;               // 0 off, 1 events + device clocks, 2 device clocks only
static bool g_open = false;          // a ProfScope is open and has not been bound to a launch yet
static std::vector<Rec> g_recs;
static unsigned long long* g_slots = nullptr;   // device [MAX_SLOTS][CLK_LINES][CLK_LINE_U64], all ones = untouched
static int g_nslots = 0;

bool prof_enabled() { return g_mode != 0; }

void prof_begin(const char* kernel, double flops, double bytes) {
    Rec r{kernel, flops, bytes, nullptr, nullptr, false, -1};
    if (g_mode == 1 && (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess)) {
        (void)hipGetLastError();
        return;
    }
    g_recs.push_back(r);
    g_open = true;
}
bool prof_take_events(hipEvent_t* e0, hipEvent_t* e1) {
    if (!g_mode || !g_open) return false;
    g_open = false;
    Rec& r = g_recs.back();
    r.launched = true;
    *e0 = r.e0; *e1 = r.e1;
    return g_mode == 1;
}
unsigned long long* prof_clock_slot() {
    if (!g_mode || !g_open || !g_slots || g_nslots >= MAX_SLOTS) return nullptr;
    Rec& r = g_recs.back();
    if (r.slot < 0) r.slot = g_nslots++;
    return g_slots + SLOT_U64 * (size_t)r.slot;
}
void prof_end() {
    if (g_open) {                    // scope closed without a launch (early return): drop the record
        if (g_recs.back().e0) { hipEventDestroy(g_recs.back().e0); hipEventDestroy(g_recs.back().e1); }
        g_recs.pop_back();
        g_open = false;
    }
}

}  // namespace refil

using namespace refil;

extern "C" int refil_profile_enable(int mode) {
    for (auto& r : g_recs)
        if (r.e0) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    (void)hipGetLastError();      // never leave a sticky error behind for the next launch check
    g_recs.clear();
    g_open = false;
    g_nslots = 0;
    g_mode = 0;
    if (mode) {
        REFIL_HIP(hipDeviceSynchronize());
        if (!g_slots) REFIL_HIP(hipMalloc((void**)&g_slots, sizeof(unsigned long long) * SLOT_U64 * MAX_SLOTS));
        REFIL_HIP(hipMemset(g_slots, 0xFF, sizeof(unsigned long long) * SLOT_U64 * MAX_SLOTS));
        REFIL_HIP(hipDeviceSynchronize());
        g_mode = mode == 2 ? 2 : 1;
    }
    return 0;
}

extern "C" int refil_profile_collect(refil_profile_entry* out, int max_entries) {
    REFIL_CHECK(out && max_entries > 0, "refil_profile_collect: bad arguments");
    REFIL_HIP(hipDeviceSynchronize());
    std::vector<unsigned long long> raw(SLOT_U64 * (size_t)(g_nslots > 0 ? g_nslots : 1));
    if (g_nslots > 0) REFIL_HIP(hipMemcpy(raw.data(), g_slots, sizeof(unsigned long long) * SLOT_U64 * g_nslots, hipMemcpyDeviceToHost));
    auto span = [&](int slot, unsigned long long& t0, unsigned long long& t1) {
        t0 = ~0ull; t1 = 0ull;
        for (int l = 0; l < CLK_LINES; ++l) {
            const unsigned long long* p = raw.data() + SLOT_U64 * (size_t)slot + (size_t)l * CLK_LINE_U64;
            if (p[0] < t0) t0 = p[0];
            if (~p[1] > t1) t1 = ~p[1];
        }
    };
    int dev = 0, khz = 100000;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) { (void)hipGetLastError(); khz = 100000; }
    std::map<std::string, refil_profile_entry> agg;
    for (auto& r : g_recs) {
        if (!r.launched) continue;
        auto& e = agg[r.name];
        if (e.launches == 0) { memset(&e, 0, sizeof(e)); strncpy(e.name, r.name, sizeof(e.name) - 1); }
        e.launches += 1; e.flops += r.flops; e.bytes += r.bytes;
        if (r.e0) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { e.total_ms += ms; e.event_launches += 1; }
            else (void)hipGetLastError();
        }
        if (r.slot >= 0) {
            unsigned long long t0, t1;
            span(r.slot, t0, t1);
            if (t0 != ~0ull && t1 > t0) { e.clock_ms += (double)(t1 - t0) / (double)khz; e.clock_launches += 1; }
        }
    }
    int n = 0;
    for (auto& kv : agg) {
        if (n >= max_entries) break;
        out[n++] = kv.second;
    }
    return n < 0 ? 0 : -n;   // negative count on success so that 0/positive keep the error convention
}
