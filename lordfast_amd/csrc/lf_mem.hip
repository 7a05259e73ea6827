/*
 * lf_mem.hip -- persistent, grow-only buffer slots: device memory per (device, slot) and pinned host memory per
 * slot.  Batches reuse them, so the steady state does no hipMalloc / hipHostMalloc / hipFree at all
 * (a multi-GB hipMalloc costs 10^2 ms; pageable PCIe copies run at a fraction of the pinned rate).
 */
#include <mutex>
#include "lf_gpu_common.h"

namespace {
struct slot_t { void *p = nullptr; size_t cap = 0; };
constexpr int MAX_DEV = 16, LANE_STRIDE = 160, MAX_LANE = 64, MAX_SLOT = MAX_LANE * LANE_STRIDE;
thread_local int t_lane = 0;
hipStream_t g_streams[MAX_DEV][MAX_LANE][16];
hipEvent_t g_events[MAX_DEV][MAX_LANE][48];
slot_t g_dev[MAX_DEV][MAX_SLOT];
slot_t g_pin[MAX_SLOT];
std::mutex g_mu;
}

/* up to 32 chunks ("lanes") may be in flight at once, spread over the devices of the process (lanes 32..63 belong to the chunks' helper threads), each driven by its own host thread: a lane owns its own
 * set of slots and streams, so nothing is shared between them but the index */
extern "C" void lfg_set_lane(int lane) { t_lane = (lane >= 0 && lane < MAX_LANE) ? lane : 0; }
extern "C" int lfg_get_lane(void) { return t_lane; }
extern "C" void *lfg_lane_stream(int device, int which)
{
    if (device < 0 || device >= MAX_DEV || which < 0 || which >= 16) return nullptr;
    std::lock_guard<std::mutex> g(g_mu);
    hipStream_t &st = g_streams[device][t_lane][which];
    if (!st) { if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&st) != hipSuccess) { lf_set_error("hipStreamCreate failed"); return nullptr; } }
    return (void *)st;
}

/* persistent events of the calling thread's lane on `device` (created on first use, never destroyed: an error path
 * cannot leak them, and nothing is created or destroyed in the steady state) */
extern "C" void *lfg_lane_event(int device, int which)
{
    if (device < 0 || device >= MAX_DEV || which < 0 || which >= 48) return nullptr;
    std::lock_guard<std::mutex> g(g_mu);
    hipEvent_t &ev = g_events[device][t_lane][which];
    if (!ev) { if (hipSetDevice(device) != hipSuccess || hipEventCreate(&ev) != hipSuccess) { lf_set_error("hipEventCreate failed"); return nullptr; } }
    return (void *)ev;
}

extern "C" void *lfg_dev_slot(int device, int slot, size_t bytes)
{
    slot += t_lane * LANE_STRIDE;
    if (device < 0 || device >= MAX_DEV || slot < 0 || slot >= MAX_SLOT) { lf_set_error("bad device slot %d/%d", device, slot); return nullptr; }
    std::lock_guard<std::mutex> g(g_mu);
    slot_t &s = g_dev[device][slot];
    if (bytes + 256 <= s.cap) return s.p;
    if (s.p) { (void)hipDeviceSynchronize(); (void)hipFree(s.p); s.p = nullptr; s.cap = 0; }
    size_t want = bytes + bytes / 4 + 4096;                       /* head-room so that similar batches do not regrow */
    if (hipMalloc(&s.p, want) != hipSuccess) {
        want = bytes + 256;
        if (hipMalloc(&s.p, want) != hipSuccess) { lf_set_error("hipMalloc of %zu bytes failed (slot %d)", want, slot); s.p = nullptr; return nullptr; }
    }
    s.cap = want;
    return s.p;
}

extern "C" void *lfg_pin_slot(int slot, size_t bytes)
{
    slot += t_lane * LANE_STRIDE;
    if (slot < 0 || slot >= MAX_SLOT) { lf_set_error("bad pinned slot %d", slot); return nullptr; }
    std::lock_guard<std::mutex> g(g_mu);
    slot_t &s = g_pin[slot];
    if (bytes + 64 <= s.cap) return s.p;
    if (s.p) { (void)hipDeviceSynchronize(); (void)hipHostFree(s.p); s.p = nullptr; s.cap = 0; }
    size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&s.p, want, hipHostMallocDefault) != hipSuccess) { lf_set_error("hipHostMalloc of %zu bytes failed (slot %d)", want, slot); s.p = nullptr; return nullptr; }
    s.cap = want;
    return s.p;
}

extern "C" void lfg_slots_release(void)
{
    std::lock_guard<std::mutex> g(g_mu);
    (void)hipDeviceSynchronize();
    for (int d = 0; d < MAX_DEV; d++) for (int k = 0; k < MAX_SLOT; k++) if (g_dev[d][k].p) { (void)hipSetDevice(d); (void)hipFree(g_dev[d][k].p); g_dev[d][k] = slot_t(); }
    for (int k = 0; k < MAX_SLOT; k++) if (g_pin[k].p) { (void)hipHostFree(g_pin[k].p); g_pin[k] = slot_t(); }
}
