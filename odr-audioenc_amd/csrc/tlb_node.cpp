// tlb_node.cpp -- node level of include/toolame_batch.h, part (3): every GPU of one host behind one handle.
//
// "contiguous stream-id blocks [g*N/G, (g+1)*N/G) per GPU; one host process (or thread) + HIP streams per GPU,
// each with its own PCM ingest and bitstream egress; no collective on the data path".  The reference has no such layer -- one
// odr-audioenc process carries one service (AudioEnc::run(), src/odr-audioenc.cpp:819-1276) -- so this is the fleet-side caller the
// north star describes ("host C++ fans thousands of independent streams ... across the 8 GPUs of one node"), written against the
// library's own C-ABI: a shard is a tlb_tick or a tlb_batch plus a thread that issues every call on it.  Nothing in here touches a
// kernel; it is partition arithmetic, a mailbox per thread, and sums.
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/toolame_batch.h"
#include "mp2_host.h"
#include "tlb_mailbox.h"
#include "tlb_plan.h"
#ifdef TLB_FAULT_INJECT
#include "tlb_debug.h"
#endif

namespace {

double now_ns()
{
    return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// One block of streams: the object that encodes it, the thread that talks to it (the mailbox: csrc/tlb_mailbox.h).
struct Shard : TlbMailbox {
    int index = 0, device = 0, first = 0, n = 0;
    tlb_tick *tick = nullptr;
    tlb_batch *batch = nullptr;
    hipStream_t stream = nullptr;                // BATCH plane: the shard's launches are ordered on it
    // counters (written by the shard's thread inside jobs, read by the node between jobs)
    long steps = 0, frames = 0;
    double busy_ns = 0, device_ms = 0;
    std::deque<double> t_submit;                 // host clock of the steps in flight
    std::deque<long> f_submit;                   // their (stream, frame) pairs
    // BATCH plane: one pair of timing events per queued encode call (recorded on the shard's stream around the launch), summed at sync:
    // device_ms is the device time of EVERY step, however many were queued per sync
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::deque<std::pair<hipEvent_t, hipEvent_t>> ev_flight;
    // health (fault isolation): a shard whose device call failed is BROKEN -- skipped by every node-wide call, its accessors answer
    // NULL / 0 -- until tlb_node_shard_restart() has made it a fresh object; the other shards never notice
    bool broken = false;
    int last_err = 0;
    long failures = 0, restarts = 0, lost_steps = 0;
    char what[TLB_NODE_WHAT_LEN] = {};
    // what the device is (filled once at creation, on the shard's thread)
    char device_name[TLB_NODE_NAME_LEN] = {}, pci[24] = {}, uuid[36] = {};
    int num_cu = 0, num_xcd = 0;
    double hbm_gb = 0;

    // called on the shard's thread right after the failing call: what HIP last complained about in this thread, then the mark
    int fail(int rc, const char *where)
    {
        const hipError_t e = hipGetLastError();
        snprintf(what, sizeof what, "%s: code %d, hip: %s", where, rc, e == hipSuccess ? "no error recorded (an argument / state error, or injected)" : hipGetErrorString(e));
        if (!broken) fprintf(stderr, "libtoolame-dab-hip: node shard %d (device %d, streams [%d, %d)) is broken -- %s; the other shards go on, tlb_node_shard_restart() brings it back\n",
                             index, device, first, first + n, what);
        broken = true; last_err = rc; failures++;
        lost_steps += (long)t_submit.size();                        // what was in flight is lost with it; the queues are emptied so that the
        t_submit.clear(); f_submit.clear();                         // counters stay those of completed steps
        for (auto &e : ev_flight) ev_pool.push_back(e);
        ev_flight.clear();
        return rc;
    }
    bool live() const { return !broken && (tick || batch); }
};

}  // namespace

struct tlb_node {
    int plane = TLB_NODE_TICK, nstreams = 0;
    tlb_node_config cfg;                         // as given at creation (the version string copied into `version`)
    std::string version, describe;
    std::vector<tlb_stream_config> cfgs;         // the CURRENT configuration of every stream (reconfigurations applied): what a restarted shard is made from
    std::vector<double> gain_db;                 // the caller's gains, re-applied to a restarted shard
    std::vector<Shard *> shards;
    std::vector<int> shard_of;
    bool finished = false;
    // node-level clock: first submit -> last wait of a step
    std::deque<double> t_submit;
    double wall_ns = 0;

    // Run fn(shard) on the thread of every LIVE shard at once.  A shard whose fn returns non-zero is marked broken there and then (on
    // its own thread, with HIP's last error of that thread) and is skipped from now on; the others are not disturbed.  Returns the
    // first non-zero code of THIS call -- the caller's cue to look at tlb_node_shard_status() -- or TLB_ERR_HIP when no shard is live.
    int live(const char *where, const std::function<int(Shard &)> &fn)
    {
        std::vector<Shard *> on;
        for (Shard *s : shards)
            if (s->live()) { on.push_back(s); s->post([s, &fn, where] { const int r = fn(*s); return r ? s->fail(r, where) : 0; }); }
        int rc = on.empty() ? (int)TLB_ERR_HIP : 0;
        for (Shard *s : on) { const int r = s->join_job(); if (r && !rc) rc = r; }
        return rc;
    }
    // every shard, broken or not (creation, teardown, the caller's own per-block work); first non-zero code wins, nothing is marked
    int all(const std::function<int(Shard &)> &fn)
    {
        for (Shard *s : shards) s->post([s, &fn] { return fn(*s); });
        int rc = 0;
        for (Shard *s : shards) { const int r = s->join_job(); if (r && !rc) rc = r; }
        return rc;
    }
    int one(int shard, const std::function<int(Shard &)> &fn)
    {
        Shard *s = shards[(size_t)shard];
        s->post([s, &fn] { return fn(*s); });
        return s->join_job();
    }
    Shard *of(int stream, int *local) const
    {
        if (stream < 0 || stream >= nstreams) return nullptr;
        Shard *s = shards[(size_t)shard_of[(size_t)stream]];
        *local = stream - s->first;
        return s;
    }
    // the owning shard of a stream if its results may be read: NULL for a broken shard (its buffers hold a half-finished step)
    Shard *read(int stream, int *local) const
    {
        Shard *s = of(stream, local);
        return s && s->live() ? s : nullptr;
    }
};

namespace {

// the objects of one shard, made on the shard's own thread (creation and restart)
int shard_make(tlb_node *nd, Shard &s, long long now_s)
{
    int e = 0;
    if (hipSetDevice(s.device) != hipSuccess) return TLB_ERR_HIP;
    if (nd->plane == TLB_NODE_TICK) {
        tlb_tick_config tc = nd->cfg.tick;
        tc.version = nd->version.data(); tc.version_len = (int)nd->version.size();
        if (now_s >= 0) tc.now_s = now_s;
        s.tick = tlb_tick_create(s.device, s.n, nd->cfgs.data() + s.first, &tc, &e);
        if (!s.tick) return e ? e : TLB_ERR_HIP;
    } else {
        s.batch = tlb_create(s.device, s.n, nd->cfgs.data() + s.first, &e);
        if (!s.batch) return e ? e : TLB_ERR_HIP;
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) return TLB_ERR_HIP;
    }
    for (int k = 0; k < s.n; k++) {                                  // the caller's gains (0 dB needs no call)
        const double g = nd->gain_db[(size_t)(s.first + k)];
        if (g == 0.0) continue;
        if (int rc = s.tick ? tlb_tick_set_gain_db(s.tick, k, g) : tlb_set_gain_db(s.batch, k, g)) return rc;
    }
    return 0;
}
void shard_unmake(Shard &s)
{
    (void)hipSetDevice(s.device);
    if (s.tick) tlb_tick_destroy(s.tick);
    if (s.batch) tlb_destroy(s.batch);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    for (auto &e : s.ev_flight) s.ev_pool.push_back(e);
    s.ev_flight.clear();
    for (auto &e : s.ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    s.ev_pool.clear();
    s.tick = nullptr; s.batch = nullptr; s.stream = nullptr;
    s.t_submit.clear(); s.f_submit.clear();
}
// what the shard's device is: so that a record of a multi-GPU run can show N DISTINCT devices took part
void shard_identify(Shard &s)
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, s.device) != hipSuccess) { snprintf(s.device_name, sizeof s.device_name, "device %d (no properties)", s.device); return; }
    snprintf(s.device_name, sizeof s.device_name, "%s%s%s", p.name, p.name[0] ? " " : "", p.gcnArchName);     // (some boxes report no marketing name: the ISA name says what it is)
    snprintf(s.pci, sizeof s.pci, "%04x:%02x:%02x.0", (unsigned)p.pciDomainID, (unsigned)p.pciBusID, (unsigned)p.pciDeviceID);
    for (int i = 0; i < 16; i++) snprintf(s.uuid + 2 * i, 3, "%02x", (unsigned)(unsigned char)p.uuid.bytes[i]);
    s.num_cu = p.multiProcessorCount;
    s.hbm_gb = (double)p.totalGlobalMem / 1e9;
    int x = 0;
    if (hipDeviceGetAttribute(&x, hipDeviceAttributeNumberOfXccs, s.device) == hipSuccess) s.num_xcd = x;
}

}  // namespace

extern "C" {

void tlb_node_partition(int nstreams, int nshards, int shard, int *first, int *n)
{
    int f = 0, c = 0;
    if (nstreams > 0 && nshards > 0 && shard >= 0 && shard < nshards) {
        f = (int)((long long)nstreams * shard / nshards);
        c = (int)((long long)nstreams * (shard + 1) / nshards) - f;
    }
    if (first) *first = f;
    if (n) *n = c;
}

int tlb_node_plan_shard(int nstreams, const tlb_stream_config *cfgs, int nshards, int shard,
                        int *first, int *n, int *nconfigs, int list_sizes[4], int *mono_pairs)
{
    if (nstreams <= 0 || !cfgs || nshards <= 0 || shard < 0 || shard >= nshards) return TLB_ERR_ARG;
    int f, c;
    tlb_node_partition(nstreams, nshards, shard, &f, &c);
    if (first) *first = f;
    if (n) *n = c;
    // what tlb_create() derives from the block -- with tlb_create's own helpers (csrc/tlb_plan.h): streams with the same knobs share a
    // record; a kernel list per psy model (4 rides with 2); mono streams of one record pair up in stream order
    std::vector<tlb_stream_config> uniq;
    std::vector<TlConfig> configs;
    std::vector<int32_t> stream_cfg, partner;
    if (int rc = tlb_plan_configs(c, cfgs + f, uniq, configs, stream_cfg)) return rc;
    const int pairs = tlb_plan_pairs(configs, stream_cfg, partner);
    int lists[4] = {0, 0, 0, 0};
    for (int s = 0; s < c; s++) lists[tlb_model_list(configs[(size_t)stream_cfg[(size_t)s]].psy)]++;
    if (nconfigs) *nconfigs = (int)uniq.size();
    if (list_sizes) for (int p = 0; p < 4; p++) list_sizes[p] = lists[p];
    if (mono_pairs) *mono_pairs = pairs;
    return TLB_OK;
}

void tlb_node_destroy(tlb_node *nd)
{
    if (!nd) return;
    for (Shard *s : nd->shards) {
        if (s->th.joinable()) {
            s->post([s] { shard_unmake(*s); return 0; });            // objects are torn down on the thread that made them
            (void)s->join_job();
            s->stop();
        }
        delete s;
    }
    delete nd;
}

tlb_node *tlb_node_create(int nshards, const int *devices, int nstreams, const tlb_stream_config *cfgs, const tlb_node_config *nc, int *err)
{
    auto fail = [&](int code) -> tlb_node * { if (err) *err = code; return nullptr; };
    if (nshards <= 0 || !devices || nstreams < nshards || !cfgs || !nc || (nc->plane != TLB_NODE_TICK && nc->plane != TLB_NODE_BATCH)) return fail(TLB_ERR_ARG);
    if (nc->plane == TLB_NODE_TICK && (nc->tick.version_len < 0 || (nc->tick.version_len && !nc->tick.version))) return fail(TLB_ERR_ARG);
    const int ndev = tlb_device_count();
    if (ndev <= 0) return fail(TLB_ERR_NO_DEVICE);
    for (int g = 0; g < nshards; g++) if (devices[g] < 0 || devices[g] >= ndev) return fail(TLB_ERR_NO_DEVICE);
    for (int g = 0; g < nshards; g++)                                 // every configuration is checked before a single byte of HBM is taken
        if (int rc = tlb_node_plan_shard(nstreams, cfgs, nshards, g, nullptr, nullptr, nullptr, nullptr, nullptr)) return fail(rc);
    tlb_node *nd = new tlb_node;
    nd->plane = nc->plane; nd->nstreams = nstreams; nd->cfg = *nc;
    if (nc->plane == TLB_NODE_TICK && nc->tick.version_len) nd->version.assign(nc->tick.version, (size_t)nc->tick.version_len);
    nd->cfg.tick.version = nullptr; nd->cfg.tick.version_len = 0;    // (the caller's pointer is not kept)
    nd->cfgs.assign(cfgs, cfgs + nstreams);
    nd->gain_db.assign((size_t)nstreams, 0.0);
    nd->shard_of.resize((size_t)nstreams);
    for (int g = 0; g < nshards; g++) {
        Shard *s = new Shard;
        s->index = g; s->device = devices[g];
        tlb_node_partition(nstreams, nshards, g, &s->first, &s->n);
        for (int k = s->first; k < s->first + s->n; k++) nd->shard_of[(size_t)k] = g;
        nd->shards.push_back(s);
        s->start([s] { (void)hipSetDevice(s->device); });            // HIP's current device is per thread; the tlb_* calls set it again themselves
    }
    const int rc = nd->all([&](Shard &s) { shard_identify(s); return shard_make(nd, s, -1); });
    if (rc) { tlb_node_destroy(nd); return fail(rc); }
    for (Shard *s : nd->shards) {
        char line[320];
        snprintf(line, sizeof line, "shard %d: device %d %s, %d CUs in %d XCDs, %.0f GB, pci %s, uuid %s, streams [%d, %d)\n",
                 s->index, s->device, s->device_name, s->num_cu, s->num_xcd, s->hbm_gb, s->pci, s->uuid, s->first, s->first + s->n);
        nd->describe += line;
    }
    if (getenv("TLB_VERBOSE")) fprintf(stderr, "libtoolame-dab-hip: node of %d shards, %d streams\n%s", nshards, nstreams, nd->describe.c_str());
    if (err) *err = TLB_OK;
    return nd;
}

int tlb_node_nshards(const tlb_node *nd) { return nd ? (int)nd->shards.size() : 0; }
int tlb_node_nstreams(const tlb_node *nd) { return nd ? nd->nstreams : 0; }
int tlb_node_shard_of(const tlb_node *nd, int stream) { return nd && stream >= 0 && stream < nd->nstreams ? nd->shard_of[(size_t)stream] : -1; }
const char *tlb_node_describe(const tlb_node *nd) { return nd ? nd->describe.c_str() : ""; }

// ---- health of one shard ----
int tlb_node_shard_status(const tlb_node *nd, int shard, tlb_node_shard_info *info)
{
    if (!nd || shard < 0 || shard >= (int)nd->shards.size()) return -TLB_ERR_ARG;
    const Shard &s = *nd->shards[(size_t)shard];
    if (info) {
        memset(info, 0, sizeof *info);
        info->shard = s.index; info->device = s.device; info->first = s.first; info->nstreams = s.n;
        info->state = s.live() ? TLB_SHARD_OK : TLB_SHARD_BROKEN; info->last_err = s.last_err;
        info->failures = s.failures; info->restarts = s.restarts; info->lost_steps = s.lost_steps;
        memcpy(info->what, s.what, sizeof info->what);
        memcpy(info->device_name, s.device_name, sizeof info->device_name);
        memcpy(info->pci, s.pci, sizeof info->pci); memcpy(info->uuid, s.uuid, sizeof info->uuid);
        info->num_cu = s.num_cu; info->num_xcd = s.num_xcd; info->hbm_gb = s.hbm_gb;
    }
    return s.live() ? TLB_SHARD_OK : TLB_SHARD_BROKEN;
}
// A fresh object for the block, made on the shard's own thread: its streams start "as a freshly started reference process" (tlb_stream_reset's
// contract, for the whole block: no history, no pending frame, psy 2/4 state zero, the EDI senders re-initialised from now_s), with
// the configurations the streams have NOW (reconfigurations since creation included) and the caller's gains.  The other shards are not
// touched.  Legal between steps only (no tick in flight / after tlb_node_sync): the restarted shard joins the lockstep at the next
// submit, and its first tick emits nothing (one frame of latency), exactly like a new node's.  Works on a healthy shard too.
int tlb_node_shard_restart(tlb_node *nd, int shard, long long now_s)
{
    if (!nd || shard < 0 || shard >= (int)nd->shards.size() || nd->finished || !nd->t_submit.empty()) return TLB_ERR_ARG;
    const int rc = nd->one(shard, [&](Shard &s) {
        shard_unmake(s);
        (void)hipGetLastError();                                     // the old failure is on record in `what`; start clean
        if (int r = shard_make(nd, s, now_s)) { shard_unmake(s); s.broken = true; s.last_err = r; snprintf(s.what, sizeof s.what, "restart: code %d", r); return r; }
        s.broken = false; s.restarts++;
        return 0;
    });
    return rc;
}

int tlb_node_counters(const tlb_node *nd, tlb_node_counter *per_shard, tlb_node_counter *total)
{
    if (!nd) return TLB_ERR_ARG;
    tlb_node_counter t;
    memset(&t, 0, sizeof t);
    t.shard = -1; t.device = -1; t.first = 0; t.nstreams = nd->nstreams; t.wall_ns = nd->wall_ns;
    for (size_t g = 0; g < nd->shards.size(); g++) {
        const Shard &s = *nd->shards[g];
        tlb_node_counter c;
        memset(&c, 0, sizeof c);
        c.shard = s.index; c.device = s.device; c.first = s.first; c.nstreams = s.n;
        c.steps = s.steps; c.frames = s.frames; c.busy_ns = s.busy_ns; c.device_ms = s.device_ms;
        if (per_shard) per_shard[g] = c;
        t.frames += c.frames;
        if (g == 0 || c.steps < t.steps) t.steps = c.steps;
        if (c.busy_ns > t.busy_ns) t.busy_ns = c.busy_ns;
        if (c.device_ms > t.device_ms) t.device_ms = c.device_ms;
    }
    if (total) *total = t;
    return TLB_OK;
}

int tlb_node_parallel(tlb_node *nd, void (*fn)(void *ctx, int shard, int first, int n), void *ctx)
{
    if (!nd || !fn) return TLB_ERR_ARG;
    return nd->all([&](Shard &s) { fn(ctx, s.index, s.first, s.n); return 0; });
}

// ---- TICK plane ----
int16_t *tlb_node_pcm(tlb_node *nd, int stream)
{
    int k; Shard *s = nd ? nd->read(stream, &k) : nullptr;
    int16_t *p = s && s->tick ? tlb_tick_pcm(s->tick) : nullptr;
    return p ? p + (size_t)k * 2 * TLB_SAMPLES_PER_FRAME : nullptr;
}
uint8_t *tlb_node_xpad(tlb_node *nd, int stream)
{
    int k; Shard *s = nd ? nd->read(stream, &k) : nullptr;
    uint8_t *p = s && s->tick ? tlb_tick_xpad(s->tick) : nullptr;
    return p ? p + (size_t)k * TLB_MAX_XPAD : nullptr;
}
int32_t *tlb_node_xpad_len(tlb_node *nd, int stream)
{
    int k; Shard *s = nd ? nd->read(stream, &k) : nullptr;
    int32_t *p = s && s->tick ? tlb_tick_xpad_len(s->tick) : nullptr;
    return p ? p + k : nullptr;
}

int tlb_node_submit(tlb_node *nd)
{
    if (!nd || nd->plane != TLB_NODE_TICK || nd->finished || nd->t_submit.size() >= 2) return TLB_ERR_ARG;
    const double t0 = now_ns();
    const int rc = nd->live("tlb_tick_submit", [&](Shard &s) {
        const double t = now_ns();
        if (int r = tlb_tick_submit(s.tick)) return r;
        s.t_submit.push_back(t); s.f_submit.push_back((long)s.n);
        return 0;
    });
    nd->t_submit.push_back(t0);                                      // the node's step exists whatever a shard did: wait() retires it
    return rc;
}
int tlb_node_wait(tlb_node *nd)
{
    if (!nd || nd->plane != TLB_NODE_TICK || nd->t_submit.empty()) return TLB_ERR_ARG;
    const int rc = nd->live("tlb_tick_wait", [&](Shard &s) {
        if (s.t_submit.empty()) return 0;                            // (cannot happen in lockstep; a shard without a tick in flight has nothing to wait for)
        if (int r = tlb_tick_wait(s.tick)) return r;
        const double t = now_ns();
        s.busy_ns += t - s.t_submit.front(); s.frames += s.f_submit.front(); s.t_submit.pop_front(); s.f_submit.pop_front();
        s.steps++;
        const float ms = tlb_tick_last_ms(s.tick);
        if (ms > 0) s.device_ms += ms;
        return 0;
    });
    nd->wall_ns += now_ns() - nd->t_submit.front(); nd->t_submit.pop_front();
    return rc;
}
int tlb_node_run(tlb_node *nd)
{
    if (!nd || !nd->t_submit.empty()) return TLB_ERR_ARG;
    const int rc = tlb_node_submit(nd);
    if (rc == TLB_ERR_ARG) return rc;
    const int rw = tlb_node_wait(nd);                                // the shards that did submit are waited for even when another broke
    return rc ? rc : rw;
}
int tlb_node_finish(tlb_node *nd)
{
    if (!nd || nd->plane != TLB_NODE_TICK || nd->finished || !nd->t_submit.empty()) return TLB_ERR_ARG;
    const int rc = nd->live("tlb_tick_finish", [&](Shard &s) { return tlb_tick_finish(s.tick); });
    nd->finished = true;
    return rc;
}
int tlb_node_units(const tlb_node *nd, int stream)
{
    int k; Shard *s = nd ? nd->read(stream, &k) : nullptr;
    return s && s->tick ? tlb_tick_units(s->tick, k) : 0;
}
const int16_t *tlb_node_peaks(const tlb_node *nd, int stream)
{
    int k; Shard *s = nd ? nd->read(stream, &k) : nullptr;
    const int16_t *p = s && s->tick ? tlb_tick_peaks(s->tick) : nullptr;
    return p ? p + 2 * (size_t)k : nullptr;
}
uint32_t tlb_node_silence_ms(const tlb_node *nd, int stream)
{
    int k; Shard *s = nd ? nd->read(stream, &k) : nullptr;
    const uint32_t *p = s && s->tick ? tlb_tick_silence_ms(s->tick) : nullptr;
    return p ? p[k] : 0;
}
const uint8_t *tlb_node_frame(const tlb_node *nd, int stream, int *len)
{
    int k; Shard *s = nd ? nd->read(stream, &k) : nullptr;
    if (!s && len) *len = 0;
    return s && s->tick ? tlb_tick_frame(s->tick, k, len) : nullptr;
}
const uint8_t *tlb_node_packet(const tlb_node *nd, int stream, int unit, int *len)
{
    int k; Shard *s = nd ? nd->read(stream, &k) : nullptr;
    if (!s && len) *len = 0;
    return s && s->tick ? tlb_tick_packet(s->tick, k, unit, len) : nullptr;
}
const uint8_t *tlb_node_message(const tlb_node *nd, int stream, int unit, int *len)
{
    int k; Shard *s = nd ? nd->read(stream, &k) : nullptr;
    if (!s && len) *len = 0;
    return s && s->tick ? tlb_tick_message(s->tick, k, unit, len) : nullptr;
}
int tlb_node_fragments(const tlb_node *nd, int stream, int unit)
{
    int k; Shard *s = nd ? nd->read(stream, &k) : nullptr;
    return s && s->tick ? tlb_tick_fragments(s->tick, k, unit) : 0;
}
const uint8_t *tlb_node_fragment(const tlb_node *nd, int stream, int unit, int kf, int *len)
{
    int k; Shard *s = nd ? nd->read(stream, &k) : nullptr;
    if (!s && len) *len = 0;
    return s && s->tick ? tlb_tick_fragment(s->tick, k, unit, kf, len) : nullptr;
}

// ---- both planes: gain, life cycle of one stream (on the owning shard's thread, like every other call on the shard's object) ----
// A stream of a BROKEN shard answers TLB_ERR_HIP (the gain is remembered and applied when the shard is restarted).
int tlb_node_set_gain_db(tlb_node *nd, int stream, double gain_db)
{
    if (!nd || stream < -1 || stream >= nd->nstreams) return TLB_ERR_ARG;
    auto f = [&](Shard &s, int k) { return s.tick ? tlb_tick_set_gain_db(s.tick, k, gain_db) : tlb_set_gain_db(s.batch, k, gain_db); };
    if (stream < 0) {
        for (double &g : nd->gain_db) g = gain_db;
        return nd->live("set_gain_db", [&](Shard &s) { return f(s, -1); });
    }
    nd->gain_db[(size_t)stream] = gain_db;
    int k; Shard *s = nd->of(stream, &k);
    if (!s->live()) return TLB_ERR_HIP;
    return nd->one(s->index, [&](Shard &sh) { return f(sh, k); });
}
int tlb_node_stream_reset(tlb_node *nd, int stream)
{
    int k; Shard *s = nd ? nd->of(stream, &k) : nullptr;
    if (!s) return TLB_ERR_ARG;
    if (!s->live()) return TLB_ERR_HIP;
    return nd->one(s->index, [&](Shard &sh) { return sh.tick ? tlb_tick_stream_reset(sh.tick, k) : tlb_stream_reset(sh.batch, k); });
}
int tlb_node_stream_finish(tlb_node *nd, int stream, uint8_t *out, size_t out_size)
{
    int k; Shard *s = nd ? nd->of(stream, &k) : nullptr;
    if (!s) return -TLB_ERR_ARG;
    if (!s->live()) return -TLB_ERR_HIP;
    return nd->one(s->index, [&](Shard &sh) { return sh.tick ? tlb_tick_stream_finish(sh.tick, k, out, out_size) : tlb_stream_finish(sh.batch, k, out, out_size); });
}
int tlb_node_stream_reconfigure(tlb_node *nd, int stream, const tlb_stream_config *cfg)
{
    int k; Shard *s = nd ? nd->of(stream, &k) : nullptr;
    if (!s || !cfg) return TLB_ERR_ARG;
    if (!s->live()) return TLB_ERR_HIP;
    const int rc = nd->one(s->index, [&](Shard &sh) { return sh.tick ? tlb_tick_stream_reconfigure(sh.tick, k, cfg) : tlb_stream_reconfigure(sh.batch, k, cfg); });
    if (!rc) nd->cfgs[(size_t)stream] = *cfg;                        // a restart of the shard re-creates the stream as it is NOW
    return rc;
}

// ---- BATCH plane ----
tlb_batch *tlb_node_batch(tlb_node *nd, int shard)
{
    return nd && shard >= 0 && shard < (int)nd->shards.size() && nd->shards[(size_t)shard]->live() ? nd->shards[(size_t)shard]->batch : nullptr;
}
void *tlb_node_device_alloc(tlb_node *nd, int shard, size_t bytes)
{
    if (!nd || shard < 0 || shard >= (int)nd->shards.size() || !bytes) return nullptr;
    void *p = nullptr;
    nd->one(shard, [&](Shard &s) {
        if (hipSetDevice(s.device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) { p = nullptr; return (int)TLB_ERR_HIP; }
        if (hipMemset(p, 0, bytes) != hipSuccess) { (void)hipFree(p); p = nullptr; return (int)TLB_ERR_HIP; }   // never a buffer that is not zeroed
        return 0;
    });
    return p;
}
void tlb_node_device_free(tlb_node *nd, int shard, void *d_ptr)
{
    if (!nd || shard < 0 || shard >= (int)nd->shards.size() || !d_ptr) return;
    nd->one(shard, [&](Shard &s) { (void)hipSetDevice(s.device); (void)hipFree(d_ptr); return 0; });
}
int tlb_node_copy_in(tlb_node *nd, int shard, void *d_dst, const void *src, size_t bytes)
{
    if (!nd || shard < 0 || shard >= (int)nd->shards.size() || !d_dst || !src) return TLB_ERR_ARG;
    return nd->one(shard, [&](Shard &s) { return hipSetDevice(s.device) == hipSuccess && hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : (int)TLB_ERR_HIP; });
}
int tlb_node_copy_out(tlb_node *nd, int shard, void *dst, const void *d_src, size_t bytes)
{
    if (!nd || shard < 0 || shard >= (int)nd->shards.size() || !dst || !d_src) return TLB_ERR_ARG;
    return nd->one(shard, [&](Shard &s) {
        if (hipSetDevice(s.device) != hipSuccess) return (int)TLB_ERR_HIP;
        if (s.stream && hipStreamSynchronize(s.stream) != hipSuccess) return (int)TLB_ERR_HIP;
        return hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : (int)TLB_ERR_HIP;
    });
}
int tlb_node_encode_device(tlb_node *nd, const int16_t *const *d_pcm, int nframes, const uint8_t *const *d_xpad,
                           const int32_t *const *d_xpad_len, uint8_t *const *d_out, int32_t *const *d_out_len)
{
    if (!nd || nd->plane != TLB_NODE_BATCH || !d_pcm || !d_out || nframes <= 0) return TLB_ERR_ARG;
    const double t0 = now_ns();
    const int rc = nd->live("tlb_encode_device_len", [&](Shard &s) {
        const int g = s.index;
        const double t = now_ns();
        if (hipSetDevice(s.device) != hipSuccess) return (int)TLB_ERR_HIP;
        std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
        if (!s.ev_pool.empty()) { ev = s.ev_pool.back(); s.ev_pool.pop_back(); }
        else if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) return (int)TLB_ERR_HIP;
        int r = hipEventRecord(ev.first, s.stream) == hipSuccess ? 0 : (int)TLB_ERR_HIP;
        if (!r) r = tlb_encode_device_len(s.batch, d_pcm[g], nframes, d_xpad ? d_xpad[g] : nullptr, d_xpad_len ? d_xpad_len[g] : nullptr,
                                          d_out[g], d_out_len ? d_out_len[g] : nullptr, s.stream);
        if (!r && hipEventRecord(ev.second, s.stream) != hipSuccess) r = (int)TLB_ERR_HIP;
        if (r) { s.ev_pool.push_back(ev); return r; }
        s.ev_flight.push_back(ev);
        s.t_submit.push_back(t); s.f_submit.push_back((long)s.n * nframes);
        return 0;
    });
    nd->t_submit.push_back(t0);
    return rc;
}
int tlb_node_flush_device(tlb_node *nd, uint8_t *const *d_out, int32_t *const *d_out_len)
{
    if (!nd || nd->plane != TLB_NODE_BATCH || !d_out) return TLB_ERR_ARG;
    return nd->live("tlb_flush_device_len", [&](Shard &s) { return tlb_flush_device_len(s.batch, d_out[s.index], d_out_len ? d_out_len[s.index] : nullptr, s.stream); });
}
int tlb_node_sync(tlb_node *nd)
{
    if (!nd || nd->plane != TLB_NODE_BATCH) return TLB_ERR_ARG;
    const int rc = nd->live("hipStreamSynchronize", [&](Shard &s) {
        if (hipSetDevice(s.device) != hipSuccess || hipStreamSynchronize(s.stream) != hipSuccess) return (int)TLB_ERR_HIP;
        const double t = now_ns();
        if (!s.t_submit.empty()) {
            // launches of one shard run in order on its stream: what is in flight is busy from the oldest submit to now
            s.busy_ns += t - s.t_submit.front();
            for (auto &e : s.ev_flight) {                           // every queued launch has its own pair of events: the sum is exact
                float ms = 0;
                if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess && ms > 0) s.device_ms += ms;
                s.ev_pool.push_back(e);
            }
            s.ev_flight.clear();
            while (!s.t_submit.empty()) { s.frames += s.f_submit.front(); s.steps++; s.t_submit.pop_front(); s.f_submit.pop_front(); }
        }
        return 0;
    });
    if (!nd->t_submit.empty()) { nd->wall_ns += now_ns() - nd->t_submit.front(); nd->t_submit.clear(); }
    return rc;
}
#ifdef TLB_FAULT_INJECT
// test builds only (csrc/tlb_debug.h): the nth launch / tick from now of ONE shard fails as a device call would
int tlb_debug_node_fail_next(tlb_node *nd, int shard, int nth)
{
    if (!nd || shard < 0 || shard >= (int)nd->shards.size()) return TLB_ERR_ARG;
    Shard &s = *nd->shards[(size_t)shard];
    return s.tick ? tlb_debug_tick_fail_next(s.tick, nth) : s.batch ? tlb_debug_fail_next(s.batch, nth) : (int)TLB_ERR_ARG;
}
#endif

}  // extern "C"
