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
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/toolame_batch.h"
#include "mp2_host.h"

namespace {

double now_ns()
{
    return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// One block of streams: the object that encodes it, the thread that talks to it.
struct Shard {
    int index = 0, device = 0, first = 0, n = 0;
    tlb_tick *tick = nullptr;
    tlb_batch *batch = nullptr;
    hipStream_t stream = nullptr;                // BATCH plane: the shard's launches are ordered on it
    // counters (written by the shard's thread inside jobs, read by the node between jobs)
    long steps = 0, frames = 0;
    double busy_ns = 0, device_ms = 0;
    std::deque<double> t_submit;                 // host clock of the steps in flight
    std::deque<long> f_submit;                   // their (stream, frame) pairs
    // mailbox
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, quit = false, done = false;
    int rc = 0;

    void loop()
    {
        (void)hipSetDevice(device);              // HIP's current device is per thread; the tlb_* calls set it again themselves
        for (;;) {
            std::function<int()> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return has_job || quit; });
                if (quit && !has_job) return;
                j = std::move(job);
                has_job = false;
            }
            const int r = j();
            {
                std::lock_guard<std::mutex> lk(mu);
                rc = r; done = true;
            }
            cv.notify_all();
        }
    }
    void post(std::function<int()> j)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = std::move(j); has_job = true; done = false;
        }
        cv.notify_all();
    }
    int join_job()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
};

}  // namespace

struct tlb_node {
    int plane = TLB_NODE_TICK, nstreams = 0;
    std::vector<Shard *> shards;
    std::vector<int> shard_of;
    // node-level clock: first submit -> last wait of a step
    std::deque<double> t_submit;
    double wall_ns = 0;

    // run fn(shard) on every shard's thread at once; first non-zero code wins
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
};

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
    // what tlb_create() derives from the block (csrc/toolame_hip.hip: tlb_create_impl, batch_build_lists): streams with the same six
    // knobs share a record; a kernel list per psy model (4 rides with 2); mono streams of one record pair up in stream order
    std::vector<tlb_stream_config> uniq;
    std::vector<int> nch, open_mono;
    int lists[4] = {0, 0, 0, 0}, pairs = 0;
    for (int s = f; s < f + c; s++) {
        int u = -1;
        for (size_t i = 0; i < uniq.size(); i++)
            if (uniq[i].samplerate == cfgs[s].samplerate && uniq[i].mode == cfgs[s].mode && uniq[i].bitrate == cfgs[s].bitrate &&
                uniq[i].psy_model == cfgs[s].psy_model && uniq[i].pad_len == cfgs[s].pad_len) { u = (int)i; break; }
        if (u < 0) {
            TlConfig *C = new TlConfig;
            const int rc = tl_build_config(C, cfgs[s].samplerate, cfgs[s].mode, cfgs[s].bitrate, cfgs[s].psy_model, cfgs[s].pad_len);
            const int ch = C->nch;
            delete C;
            if (rc) return rc;
            uniq.push_back(cfgs[s]); nch.push_back(ch); open_mono.push_back(0);
            u = (int)uniq.size() - 1;
        }
        const int m = cfgs[s].psy_model == 4 ? 2 : cfgs[s].psy_model;
        lists[m]++;
        if (nch[(size_t)u] == 1) { if (open_mono[(size_t)u]) { pairs++; open_mono[(size_t)u] = 0; } else open_mono[(size_t)u] = 1; }
    }
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
            s->post([s] {                                            // objects are torn down on the thread that made them
                if (s->tick) tlb_tick_destroy(s->tick);
                if (s->batch) tlb_destroy(s->batch);
                if (s->stream) (void)hipStreamDestroy(s->stream);
                s->tick = nullptr; s->batch = nullptr; s->stream = nullptr;
                return 0;
            });
            (void)s->join_job();
            { std::lock_guard<std::mutex> lk(s->mu); s->quit = true; }
            s->cv.notify_all();
            s->th.join();
        }
        delete s;
    }
    delete nd;
}

tlb_node *tlb_node_create(int nshards, const int *devices, int nstreams, const tlb_stream_config *cfgs, const tlb_node_config *nc, int *err)
{
    auto fail = [&](int code) -> tlb_node * { if (err) *err = code; return nullptr; };
    if (nshards <= 0 || !devices || nstreams < nshards || !cfgs || !nc || (nc->plane != TLB_NODE_TICK && nc->plane != TLB_NODE_BATCH)) return fail(TLB_ERR_ARG);
    const int ndev = tlb_device_count();
    if (ndev <= 0) return fail(TLB_ERR_NO_DEVICE);
    for (int g = 0; g < nshards; g++) if (devices[g] < 0 || devices[g] >= ndev) return fail(TLB_ERR_NO_DEVICE);
    for (int g = 0; g < nshards; g++)                                 // every configuration is checked before a single byte of HBM is taken
        if (int rc = tlb_node_plan_shard(nstreams, cfgs, nshards, g, nullptr, nullptr, nullptr, nullptr, nullptr)) return fail(rc);
    tlb_node *nd = new tlb_node;
    nd->plane = nc->plane; nd->nstreams = nstreams;
    nd->shard_of.resize((size_t)nstreams);
    for (int g = 0; g < nshards; g++) {
        Shard *s = new Shard;
        s->index = g; s->device = devices[g];
        tlb_node_partition(nstreams, nshards, g, &s->first, &s->n);
        for (int k = s->first; k < s->first + s->n; k++) nd->shard_of[(size_t)k] = g;
        nd->shards.push_back(s);
        s->th = std::thread([s] { s->loop(); });
    }
    const tlb_node_config cfg = *nc;
    const int rc = nd->all([&](Shard &s) {
        int e = 0;
        if (cfg.plane == TLB_NODE_TICK) {
            s.tick = tlb_tick_create(s.device, s.n, cfgs + s.first, &cfg.tick, &e);
            return s.tick ? 0 : (e ? e : TLB_ERR_HIP);
        }
        s.batch = tlb_create(s.device, s.n, cfgs + s.first, &e);
        if (!s.batch) return e ? e : TLB_ERR_HIP;
        if (hipSetDevice(s.device) != hipSuccess || hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) return (int)TLB_ERR_HIP;
        return 0;
    });
    if (rc) { tlb_node_destroy(nd); return fail(rc); }
    if (err) *err = TLB_OK;
    return nd;
}

int tlb_node_nshards(const tlb_node *nd) { return nd ? (int)nd->shards.size() : 0; }
int tlb_node_nstreams(const tlb_node *nd) { return nd ? nd->nstreams : 0; }
int tlb_node_shard_of(const tlb_node *nd, int stream) { return nd && stream >= 0 && stream < nd->nstreams ? nd->shard_of[(size_t)stream] : -1; }

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
    int k; Shard *s = nd ? nd->of(stream, &k) : nullptr;
    int16_t *p = s && s->tick ? tlb_tick_pcm(s->tick) : nullptr;
    return p ? p + (size_t)k * 2 * TLB_SAMPLES_PER_FRAME : nullptr;
}
uint8_t *tlb_node_xpad(tlb_node *nd, int stream)
{
    int k; Shard *s = nd ? nd->of(stream, &k) : nullptr;
    uint8_t *p = s && s->tick ? tlb_tick_xpad(s->tick) : nullptr;
    return p ? p + (size_t)k * TLB_MAX_XPAD : nullptr;
}
int32_t *tlb_node_xpad_len(tlb_node *nd, int stream)
{
    int k; Shard *s = nd ? nd->of(stream, &k) : nullptr;
    int32_t *p = s && s->tick ? tlb_tick_xpad_len(s->tick) : nullptr;
    return p ? p + k : nullptr;
}

int tlb_node_submit(tlb_node *nd)
{
    if (!nd || nd->plane != TLB_NODE_TICK) return TLB_ERR_ARG;
    const double t0 = now_ns();
    const int rc = nd->all([&](Shard &s) {
        const double t = now_ns();
        if (int r = tlb_tick_submit(s.tick)) return r;
        s.t_submit.push_back(t); s.f_submit.push_back((long)s.n);
        return 0;
    });
    if (!rc) nd->t_submit.push_back(t0);
    return rc;
}
int tlb_node_wait(tlb_node *nd)
{
    if (!nd || nd->plane != TLB_NODE_TICK) return TLB_ERR_ARG;
    const int rc = nd->all([&](Shard &s) {
        if (int r = tlb_tick_wait(s.tick)) return r;
        const double t = now_ns();
        if (!s.t_submit.empty()) { s.busy_ns += t - s.t_submit.front(); s.frames += s.f_submit.front(); s.t_submit.pop_front(); s.f_submit.pop_front(); }
        s.steps++;
        const float ms = tlb_tick_last_ms(s.tick);
        if (ms > 0) s.device_ms += ms;
        return 0;
    });
    if (!rc && !nd->t_submit.empty()) { nd->wall_ns += now_ns() - nd->t_submit.front(); nd->t_submit.pop_front(); }
    return rc;
}
int tlb_node_run(tlb_node *nd)
{
    if (int rc = tlb_node_submit(nd)) return rc;
    return tlb_node_wait(nd);
}
int tlb_node_finish(tlb_node *nd)
{
    if (!nd || nd->plane != TLB_NODE_TICK) return TLB_ERR_ARG;
    return nd->all([&](Shard &s) { return tlb_tick_finish(s.tick); });
}
int tlb_node_units(const tlb_node *nd, int stream)
{
    int k; Shard *s = nd ? nd->of(stream, &k) : nullptr;
    return s && s->tick ? tlb_tick_units(s->tick, k) : 0;
}
const int16_t *tlb_node_peaks(const tlb_node *nd, int stream)
{
    int k; Shard *s = nd ? nd->of(stream, &k) : nullptr;
    const int16_t *p = s && s->tick ? tlb_tick_peaks(s->tick) : nullptr;
    return p ? p + 2 * (size_t)k : nullptr;
}
uint32_t tlb_node_silence_ms(const tlb_node *nd, int stream)
{
    int k; Shard *s = nd ? nd->of(stream, &k) : nullptr;
    const uint32_t *p = s && s->tick ? tlb_tick_silence_ms(s->tick) : nullptr;
    return p ? p[k] : 0;
}
const uint8_t *tlb_node_frame(const tlb_node *nd, int stream, int *len)
{
    int k; Shard *s = nd ? nd->of(stream, &k) : nullptr;
    return s && s->tick ? tlb_tick_frame(s->tick, k, len) : nullptr;
}
const uint8_t *tlb_node_packet(const tlb_node *nd, int stream, int unit, int *len)
{
    int k; Shard *s = nd ? nd->of(stream, &k) : nullptr;
    return s && s->tick ? tlb_tick_packet(s->tick, k, unit, len) : nullptr;
}
const uint8_t *tlb_node_message(const tlb_node *nd, int stream, int unit, int *len)
{
    int k; Shard *s = nd ? nd->of(stream, &k) : nullptr;
    return s && s->tick ? tlb_tick_message(s->tick, k, unit, len) : nullptr;
}
int tlb_node_fragments(const tlb_node *nd, int stream, int unit)
{
    int k; Shard *s = nd ? nd->of(stream, &k) : nullptr;
    return s && s->tick ? tlb_tick_fragments(s->tick, k, unit) : 0;
}
const uint8_t *tlb_node_fragment(const tlb_node *nd, int stream, int unit, int kf, int *len)
{
    int k; Shard *s = nd ? nd->of(stream, &k) : nullptr;
    return s && s->tick ? tlb_tick_fragment(s->tick, k, unit, kf, len) : nullptr;
}

// ---- both planes: gain, life cycle of one stream (on the owning shard's thread, like every other call on the shard's object) ----
int tlb_node_set_gain_db(tlb_node *nd, int stream, double gain_db)
{
    if (!nd || stream < -1 || stream >= nd->nstreams) return TLB_ERR_ARG;
    auto f = [&](Shard &s, int k) { return s.tick ? tlb_tick_set_gain_db(s.tick, k, gain_db) : tlb_set_gain_db(s.batch, k, gain_db); };
    if (stream < 0) return nd->all([&](Shard &s) { return f(s, -1); });
    int k; Shard *s = nd->of(stream, &k);
    return nd->one(s->index, [&](Shard &sh) { return f(sh, k); });
}
int tlb_node_stream_reset(tlb_node *nd, int stream)
{
    int k; Shard *s = nd ? nd->of(stream, &k) : nullptr;
    if (!s) return TLB_ERR_ARG;
    return nd->one(s->index, [&](Shard &sh) { return sh.tick ? tlb_tick_stream_reset(sh.tick, k) : tlb_stream_reset(sh.batch, k); });
}
int tlb_node_stream_finish(tlb_node *nd, int stream, uint8_t *out, size_t out_size)
{
    int k; Shard *s = nd ? nd->of(stream, &k) : nullptr;
    if (!s) return -TLB_ERR_ARG;
    return nd->one(s->index, [&](Shard &sh) { return sh.tick ? tlb_tick_stream_finish(sh.tick, k, out, out_size) : tlb_stream_finish(sh.batch, k, out, out_size); });
}
int tlb_node_stream_reconfigure(tlb_node *nd, int stream, const tlb_stream_config *cfg)
{
    int k; Shard *s = nd ? nd->of(stream, &k) : nullptr;
    if (!s || !cfg) return TLB_ERR_ARG;
    return nd->one(s->index, [&](Shard &sh) { return sh.tick ? tlb_tick_stream_reconfigure(sh.tick, k, cfg) : tlb_stream_reconfigure(sh.batch, k, cfg); });
}

// ---- BATCH plane ----
tlb_batch *tlb_node_batch(tlb_node *nd, int shard)
{
    return nd && shard >= 0 && shard < (int)nd->shards.size() ? nd->shards[(size_t)shard]->batch : nullptr;
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
        return hipSetDevice(s.device) == hipSuccess && hipStreamSynchronize(s.stream) == hipSuccess && hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : (int)TLB_ERR_HIP;
    });
}
int tlb_node_encode_device(tlb_node *nd, const int16_t *const *d_pcm, int nframes, const uint8_t *const *d_xpad,
                           const int32_t *const *d_xpad_len, uint8_t *const *d_out, int32_t *const *d_out_len)
{
    if (!nd || nd->plane != TLB_NODE_BATCH || !d_pcm || !d_out || nframes <= 0) return TLB_ERR_ARG;
    const double t0 = now_ns();
    const int rc = nd->all([&](Shard &s) {
        const int g = s.index;
        const double t = now_ns();
        if (int r = tlb_encode_device_len(s.batch, d_pcm[g], nframes, d_xpad ? d_xpad[g] : nullptr, d_xpad_len ? d_xpad_len[g] : nullptr,
                                          d_out[g], d_out_len ? d_out_len[g] : nullptr, s.stream)) return r;
        s.t_submit.push_back(t); s.f_submit.push_back((long)s.n * nframes);
        return 0;
    });
    if (!rc) nd->t_submit.push_back(t0);
    return rc;
}
int tlb_node_flush_device(tlb_node *nd, uint8_t *const *d_out, int32_t *const *d_out_len)
{
    if (!nd || nd->plane != TLB_NODE_BATCH || !d_out) return TLB_ERR_ARG;
    return nd->all([&](Shard &s) { return tlb_flush_device_len(s.batch, d_out[s.index], d_out_len ? d_out_len[s.index] : nullptr, s.stream); });
}
int tlb_node_sync(tlb_node *nd)
{
    if (!nd || nd->plane != TLB_NODE_BATCH) return TLB_ERR_ARG;
    const int rc = nd->all([&](Shard &s) {
        if (hipSetDevice(s.device) != hipSuccess || hipStreamSynchronize(s.stream) != hipSuccess) return (int)TLB_ERR_HIP;
        const double t = now_ns();
        if (!s.t_submit.empty()) {
            // launches of one shard run in order on its stream: what is in flight is busy from the oldest submit to now
            s.busy_ns += t - s.t_submit.front();
            const float ms = tlb_last_kernel_ms(s.batch);          // the most recent launch; with several queued a lower bound
            if (ms > 0) s.device_ms += ms;
            while (!s.t_submit.empty()) { s.frames += s.f_submit.front(); s.steps++; s.t_submit.pop_front(); s.f_submit.pop_front(); }
        }
        return 0;
    });
    if (!rc && !nd->t_submit.empty()) { nd->wall_ns += now_ns() - nd->t_submit.front(); nd->t_submit.clear(); }
    return rc;
}

}  // extern "C"
