/*
 * lf_mem.hip -- persistent, grow-only buffer slots: device memory per (device, slot) and pinned host memory per
 * slot.  Batches reuse them, so the steady state does no hipMalloc / hipHostMalloc / hipFree at all
 * (a multi-GB hipMalloc costs 10^2 ms; pageable PCIe copies run at a fraction of the pinned rate).
 */
#include <mutex>
#include <time.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#define LF_NO_SYNC_WRAP 1      /* this file hands the blocking wait events out and drains streams of other devices: plain waits here */
#include "lf_gpu_common.h"
#include <vector>

namespace {
struct slot_t { void *p = nullptr; size_t cap = 0; };
constexpr int MAX_DEV = 16, LANE_STRIDE = 176, MAX_LANE = 64, MAX_SLOT = MAX_LANE * LANE_STRIDE;
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
static const char *volatile g_ph_file[MAX_LANE]; static volatile int g_ph_line[MAX_LANE];
extern "C" void lfg_phase(const char *file, int line) { g_ph_file[t_lane] = file; g_ph_line[t_lane] = line; }
extern "C" void lfg_phase_dump(void)
{
    for (int l = 0; l < MAX_LANE; l++) if (g_ph_file[l]) fprintf(stderr, "[lf watchdog] lane %d: last HIP call at %s:%d\n", l, g_ph_file[l], g_ph_line[l]);
}
/* Waits for every lane stream of `device`, one hipStreamSynchronize each.  Called before anything that makes the runtime
 * synchronise the whole device by itself (hipFree, hipMalloc of a regrown slot, index load / release): with the ~10^2
 * streams of eight lanes in use, ROCm 7.2's device-wide wait can fail to arm its signal handlers
 * ("hsa_amd_signal_async_handler() failed to set the handler") and then never returns; after explicit per-stream waits it
 * finds nothing pending.  g_mu must be held by the caller (or no lane may be running). */
static void quiesce_locked(int device)
{
    if (device < 0 || device >= MAX_DEV) return;
    for (int l = 0; l < MAX_LANE; l++) for (int k = 0; k < 16; k++) if (g_streams[device][l][k]) (void)hipStreamSynchronize(g_streams[device][l][k]);
}
extern "C" void lfg_quiesce(int device)
{
    std::lock_guard<std::mutex> g(g_mu);
    if (hipSetDevice(device) != hipSuccess) return;
    quiesce_locked(device);
}
/* LF_WATCHDOG: before device memory is released every lane stream must drain; one that does not within the limit is
 * named (lane, stream index) and the process aborts instead of hanging inside hipFree */
extern "C" void lfg_drain_check(int device)
{
    const char *w = lf_env("LF_WATCHDOG");
    if (!w || atoi(w) <= 0 || device < 0 || device >= MAX_DEV) return;
    const int limit_ms = atoi(w) * 1000;
    for (int waited = 0;; waited += 20) {
        int busy = 0;
        { std::lock_guard<std::mutex> g(g_mu);
          for (int l = 0; l < MAX_LANE; l++) for (int k = 0; k < 16; k++) if (g_streams[device][l][k] && hipStreamQuery(g_streams[device][l][k]) == hipErrorNotReady) busy++; }
        if (!busy) return;
        if (waited >= limit_ms) {
            std::lock_guard<std::mutex> g(g_mu);
            for (int l = 0; l < MAX_LANE; l++) for (int k = 0; k < 16; k++) if (g_streams[device][l][k] && hipStreamQuery(g_streams[device][l][k]) == hipErrorNotReady)
                fprintf(stderr, "[lf watchdog] device %d lane %d stream %d still busy after %d s\n", device, l, k, limit_ms / 1000);
            lfg_phase_dump(); fflush(stderr); abort();
        }
        struct timespec ts = { 0, 20 * 1000000 }; nanosleep(&ts, nullptr);
    }
}
/* At process exit the HIP runtime tears its streams down by itself; with the ~10^2 streams of the lanes still alive that
 * teardown has been seen to stall for tens of seconds or for good (ROCm 7.2).  Registered once, when the first lane stream
 * is created -- i.e. after the runtime's own handlers, so it runs BEFORE them: every lane stream is drained and destroyed,
 * events are destroyed, slots released, while the runtime is still whole. */
static void lf_exit_cleanup(void)
{
    std::lock_guard<std::mutex> g(g_mu);
    for (int d = 0; d < MAX_DEV; d++) {
        bool any = false;
        for (int l = 0; l < MAX_LANE && !any; l++) for (int k = 0; k < 16; k++) if (g_streams[d][l][k]) { any = true; break; }
        if (!any) continue;
        if (hipSetDevice(d) != hipSuccess) continue;
        quiesce_locked(d);
        for (int l = 0; l < MAX_LANE; l++) {
            for (int k = 0; k < 48; k++) if (g_events[d][l][k]) { (void)hipEventDestroy(g_events[d][l][k]); g_events[d][l][k] = nullptr; }
            for (int k = 0; k < 16; k++) if (g_streams[d][l][k]) { (void)hipStreamDestroy(g_streams[d][l][k]); g_streams[d][l][k] = nullptr; }
        }
    }
}
/* ---- blocking waits.  A host thread that waits for a stream SLEEPS on an event created with hipEventBlockingSync, whatever the
 * device's scheduling flags are: hipSetDeviceFlags(hipDeviceScheduleBlockingSync) (lfg_index_upload) is refused once the process
 * has an active context -- a host program that initialised the runtime first, like bench.py's torch -- and HIP's default wait
 * spins: eight lane drivers waiting for their chunks burnt 0.2 - 1.5 core-seconds per step that way.  One event per host thread
 * and device (two threads on one event would wait for each other's marker); a thread that ends gives its events back. ---- */
static std::mutex g_wev_mu;
static std::vector<hipEvent_t> g_wev_pool[MAX_DEV];
struct lf_wev_holder {
    hipEvent_t ev[MAX_DEV];
    lf_wev_holder() { for (int d = 0; d < MAX_DEV; d++) ev[d] = nullptr; }
    ~lf_wev_holder() { std::lock_guard<std::mutex> g(g_wev_mu); for (int d = 0; d < MAX_DEV; d++) if (ev[d]) g_wev_pool[d].push_back(ev[d]); }
};
static thread_local lf_wev_holder t_wev;
/* host waits of the calling thread since it last asked (lf_stats_t.n_host_waits: a chunk's launch chain is judged by them) */
static thread_local uint64_t t_waits = 0;
extern "C" void lfg_count_wait(void) { t_waits++; }
extern "C" uint64_t lfg_take_waits(void) { const uint64_t w = t_waits; t_waits = 0; return w; }
/* LF_WAIT_TRACE=1: a census of the waits by call site, printed when the process ends (measurement aid, profiles/tools/r05_waits.sh) */
static std::mutex g_wsite_mu;
static struct { const char *file; int line; uint64_t n; } g_wsite[128];
static int g_nwsite = 0;
static void wsite_dump(void)
{
    for (int i = 0; i < g_nwsite; i++) { const char *b = strrchr(g_wsite[i].file, '/'); fprintf(stderr, "[lf] waits %8llu at %s:%d\n", (unsigned long long)g_wsite[i].n, b ? b + 1 : g_wsite[i].file, g_wsite[i].line); }
}
extern "C" void lfg_count_wait_at(const char *file, int line)
{
    std::lock_guard<std::mutex> g(g_wsite_mu);
    for (int i = 0; i < g_nwsite; i++) if (g_wsite[i].line == line && g_wsite[i].file == file) { g_wsite[i].n++; return; }
    if (g_nwsite == 0) atexit(wsite_dump);
    if (g_nwsite < 128) { g_wsite[g_nwsite].file = file; g_wsite[g_nwsite].line = line; g_wsite[g_nwsite].n = 1; g_nwsite++; }
}
extern "C" void *lfg_thread_wait_event(int device)
{
    if (device < 0 || device >= MAX_DEV) return nullptr;
    hipEvent_t &e = t_wev.ev[device];
    if (!e) {
        { std::lock_guard<std::mutex> g(g_wev_mu); if (!g_wev_pool[device].empty()) { e = g_wev_pool[device].back(); g_wev_pool[device].pop_back(); } }
        if (!e && hipEventCreateWithFlags(&e, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); e = nullptr; }
    }
    return (void *)e;
}

extern "C" void *lfg_lane_stream(int device, int which)
{
    if (device < 0 || device >= MAX_DEV || which < 0 || which >= 16) return nullptr;
    std::lock_guard<std::mutex> g(g_mu);
    static bool registered = false;
    if (!registered) { registered = true; atexit(lf_exit_cleanup); }
    hipStream_t &st = g_streams[device][t_lane][which];
    if (!st) {
        if (hipSetDevice(device) != hipSuccess) { lf_set_error("hipSetDevice failed"); return nullptr; }
        /* (CU-masked streams -- seed / vote on every k-th CU, everything else on the rest -- were measured in round 4, profiles/r04_cu_split/: 366 ms per step
         * against 77; not in the tree) */
        const hipError_t rc = hipStreamCreate(&st);
        if (rc != hipSuccess) { lf_set_error("hipStreamCreate failed"); st = nullptr; return nullptr; }
    }
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

/* a few numbers a lane leaves for its later stages (e.g. the size of the bit planes lfg_seed made of the resident batch) */
static uint64_t g_lane_val[MAX_DEV][MAX_LANE][8];
extern "C" void lfg_lane_set_value(int device, int key, uint64_t v) { if (device >= 0 && device < MAX_DEV && key >= 0 && key < 8) g_lane_val[device][t_lane][key] = v; }
extern "C" uint64_t lfg_lane_value(int device, int key) { return (device >= 0 && device < MAX_DEV && key >= 0 && key < 8) ? g_lane_val[device][t_lane][key] : 0; }

extern "C" void *lfg_dev_slot(int device, int slot, size_t bytes)
{
    slot += t_lane * LANE_STRIDE;
    if (device < 0 || device >= MAX_DEV || slot < 0 || slot >= MAX_SLOT) { lf_set_error("bad device slot %d/%d", device, slot); return nullptr; }
    std::lock_guard<std::mutex> g(g_mu);
    slot_t &s = g_dev[device][slot];
    if (bytes + 256 <= s.cap) return s.p;
    if (s.p) { quiesce_locked(device); (void)hipFree(s.p); s.p = nullptr; s.cap = 0; }
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
    if (s.p) { for (int d = 0; d < MAX_DEV; d++) quiesce_locked(d); (void)hipHostFree(s.p); s.p = nullptr; s.cap = 0; }
    size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&s.p, want, hipHostMallocDefault) != hipSuccess) { lf_set_error("hipHostMalloc of %zu bytes failed (slot %d)", want, slot); s.p = nullptr; return nullptr; }
    s.cap = want;
    return s.p;
}

extern "C" void lfg_slots_release(void)
{
    std::lock_guard<std::mutex> g(g_mu);
    for (int d = 0; d < MAX_DEV; d++) quiesce_locked(d);
    for (int d = 0; d < MAX_DEV; d++) for (int k = 0; k < MAX_SLOT; k++) if (g_dev[d][k].p) { (void)hipSetDevice(d); (void)hipFree(g_dev[d][k].p); g_dev[d][k] = slot_t(); }
    for (int k = 0; k < MAX_SLOT; k++) if (g_pin[k].p) { (void)hipHostFree(g_pin[k].p); g_pin[k] = slot_t(); }
}

/* ---- device memory for callers of the device-resident entry points (lf_map_batch_dev) that do not link HIP themselves.
 * Plain allocations outside the slot system: the caller owns them. ---- */
extern "C" void *lf_device_alloc(int device, size_t bytes)
{
    void *p = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { lf_set_error("lf_device_alloc: hipMalloc of %zu bytes on device %d failed", bytes, device); return nullptr; }
    return p;
}
extern "C" void lf_device_free(int device, void *p)
{
    if (!p || hipSetDevice(device) != hipSuccess) return;
    lfg_quiesce(device);              /* hipFree synchronises the device by itself: per-stream waits first (see lfg_quiesce) */
    (void)hipFree(p);
}
/* dst / src: host or device memory of `device` (the runtime tells them apart); synchronous */
extern "C" int lf_device_copy(int device, void *dst, const void *src, size_t bytes)
{
    HIPCHK(hipSetDevice(device));
    lfg_quiesce(device);              /* a synchronous copy waits for the device by itself: per-stream waits first (see lfg_quiesce) */
    if (bytes) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDefault));
    return LF_OK;
}

/* pinned host memory outside the slot system (the SAM buffers of lf_map_file): D2H copies run at link speed into it */
extern "C" void *lfg_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void lfg_host_free(void *p) { if (p) (void)hipHostFree(p); }

/* ---- workspaces of the single-launch scans (lf_scan.h): status words + tile counter, per (device, lane, module) ---- */
#include "lf_scan.h"
namespace { struct scan_state_t { void *p = nullptr; size_t cap = 0; unsigned epoch = 0; }; scan_state_t g_scan[MAX_DEV][MAX_LANE][8]; }
extern "C" int lfg_scan_ws(int device, int which, size_t n, void *stream, lf_scan_ws_t *ws)
{
    if (device < 0 || device >= MAX_DEV || which < 0 || which >= 8) { lf_set_error("lfg_scan_ws: bad workspace %d/%d", device, which); return LF_ERR_ARG; }
    const size_t tiles = (n + LF_SCAN_TILE - 1) / LF_SCAN_TILE + 1;
    void *p = lfg_dev_slot(device, LF_DS_SCAN0 + which, tiles * 8 + 64);
    if (!p) return LF_ERR_NOMEM;
    scan_state_t &S = g_scan[device][t_lane][which];
    S.epoch = (S.epoch + 1) & 0x3fffu;
    size_t cap = 0;
    { std::lock_guard<std::mutex> g(g_mu); cap = g_dev[device][LF_DS_SCAN0 + which + t_lane * LANE_STRIDE].cap; }
    if (p != S.p || cap != S.cap || S.epoch == 0) {
        /* a new (regrown) array, or the epoch wrapped: stale words could carry a live epoch */
        if (hipMemsetAsync(p, 0, cap, (hipStream_t)stream) != hipSuccess) { lf_set_error("lfg_scan_ws: memset failed"); return LF_ERR_HIP; }
        S.p = p; S.cap = cap; S.epoch = 1;
    }
    ws->counter = (unsigned int *)p;                         /* word 0: the tile counter (left at 0 by every launch) */
    ws->status = (unsigned long long *)p + 1;
    ws->epoch = S.epoch;
    return LF_OK;
}
