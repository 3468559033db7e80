// nodetick -- a FLEET of services on one host: what AudioEnc::run() (src/odr-audioenc.cpp:819-1276) does for one service, done for
// N streams spread over the machine's GPUs through the node level of the C-ABI (include/toolame_batch.h part 3, tlb_node_*).
// Host code only (plain C++); the partition, the per-GPU threads and objects, and the counters live in libtoolame_dab_hip.so.
//
//   per tick (24 ms of audio):                                          reference, one service            this program, N services
//     1. every service's PCM into its slot of the pinned input set      inputs -> queue (:904-986)        tlb_node_parallel(fill)  (one thread per GPU block)
//     2. gain / peak / de-interleave, encode, re-frame, EDI AF packets  :1030-1051,1139-1163,1208-1225    tlb_node_submit / tlb_node_wait
//                                                                       + Outputs.cpp:194-261
//     3. ship the packets                                               EDI::write_frame -> sender        tlb_node_parallel(ship)  (here: hash + count)
//   two ticks are kept in flight: while tick t's packets are shipped, tick t+1 is on the GPUs and tick t+2's PCM is being filled.
//
// build: g++ -O2 -std=c++17 examples/nodetick.cpp -Iinclude -Lodr-audioenc_amd -ltoolame_dab_hip -Wl,-rpath,$PWD/odr-audioenc_amd -o nodetick
// usage: nodetick in.s16le [-n streams] [-G shards] [-d dev,dev,...] [-k ticks] [-b kbps] [-p psy] [-o out.af]
//   in.s16le: interleaved stereo 48 kHz; stream s starts reading at frame s (so the services differ), wrapping around.
//   -d: HIP device of each shard (default 0,1,...,G-1 modulo the device count; "0,0" = two shards on one GPU).
//   -o: the AF packets of the LAST stream of the node, length-prefixed (uint32 LE) -- the stream farthest from shard 0.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "toolame_batch.h"

static void die(const char *what, int code)
{
    std::fprintf(stderr, "nodetick: %s (code %d)\n", what, code);
    std::exit(1);
}

struct Ctx {
    tlb_node *nd;
    const std::vector<int16_t> *pcm;                         // the whole input file
    size_t nframes_in;
    long tick;
    std::vector<uint64_t> *hash;                             // per shard: FNV-1a over every packet byte shipped
    std::vector<long> *packets, *bytes;
};

// step 1 on shard `g`'s thread: the block's services copy their frame of this tick into the pinned input set
static void fill(void *vctx, int g, int first, int n)
{
    Ctx &c = *(Ctx *)vctx;
    for (int s = first; s < first + n; s++) {
        int16_t *dst = tlb_node_pcm(c.nd, s);
        if (!dst) {                                              // a BROKEN shard takes no input (its block is off air until it is restarted); anything else is a bug
            if (tlb_node_shard_status(c.nd, g, nullptr) == TLB_SHARD_BROKEN) return;
            die("no input set free", s);
        }
        const size_t f = ((size_t)s + (size_t)c.tick) % c.nframes_in;
        std::memcpy(dst, c.pcm->data() + f * 2304, 2304 * sizeof(int16_t));
    }
    (void)g;
}

// step 3 on shard `g`'s thread: a real sender would write each packet to its service's EDI destination
static void ship(void *vctx, int g, int first, int n)
{
    Ctx &c = *(Ctx *)vctx;
    uint64_t h = (*c.hash)[(size_t)g];
    for (int s = first; s < first + n; s++)
        for (int u = 0; u < tlb_node_units(c.nd, s); u++) {
            int len = 0;
            const uint8_t *p = tlb_node_packet(c.nd, s, u, &len);
            if (!p || !len) continue;
            for (int i = 0; i < len; i++) h = (h ^ p[i]) * 1099511628211ull;
            (*c.packets)[(size_t)g]++;
            (*c.bytes)[(size_t)g] += len;
        }
    (*c.hash)[(size_t)g] = h;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s in.s16le [-n streams] [-G shards] [-d dev,dev,...] [-k ticks] [-b kbps] [-p psy] [-o out.af]\n", argv[0]);
        return 2;
    }
    int nstreams = 64, G = 0, ticks = 50, kbps = 128, psy = 1;
    std::string devs, outpath;
    for (int i = 2; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        const char *v = argv[i + 1];
        if (k == "-n") nstreams = std::atoi(v);
        else if (k == "-G") G = std::atoi(v);
        else if (k == "-d") devs = v;
        else if (k == "-k") ticks = std::atoi(v);
        else if (k == "-b") kbps = std::atoi(v);
        else if (k == "-p") psy = std::atoi(v);
        else if (k == "-o") outpath = v;
        else die("unknown option", 0);
    }
    const int ndev = tlb_device_count();
    if (ndev <= 0) die("no GPU", ndev);
    std::vector<int> devices;
    for (size_t p = 0; p < devs.size();) {
        devices.push_back(std::atoi(devs.c_str() + p));
        p = devs.find(',', p);
        if (p == std::string::npos) break;
        p++;
    }
    if (devices.empty()) { if (G <= 0) G = ndev; for (int g = 0; g < G; g++) devices.push_back(g % ndev); }
    G = (int)devices.size();

    std::vector<int16_t> pcm;
    {
        std::FILE *fi = std::fopen(argv[1], "rb");
        if (!fi) die("cannot open input", 0);
        int16_t buf[2304];
        while (std::fread(buf, sizeof(int16_t), 2304, fi) == 2304) pcm.insert(pcm.end(), buf, buf + 2304);
        std::fclose(fi);
    }
    const size_t nframes_in = pcm.size() / 2304;
    if (!nframes_in) die("input shorter than one frame", 0);

    // the fleet: every service 48 kHz joint stereo (odr-audioenc's default mode, src/odr-audioenc.cpp:697-709)
    std::vector<tlb_stream_config> cfg((size_t)nstreams, tlb_stream_config{48000, 'j', kbps, psy, 0});
    static const char version[] = "nodetick example";
    tlb_node_config nc;
    std::memset(&nc, 0, sizeof nc);
    nc.plane = TLB_NODE_TICK;
    nc.tick.egress = TLB_TICK_EDI_AF;
    nc.tick.version = version; nc.tick.version_len = (int)std::strlen(version);
    nc.tick.now_s = 1712345678; nc.tick.tist = 1; nc.tick.tai_utc_offset = 37;
    for (int g = 0; g < G; g++) {                                // what each GPU will hold, before any of them is touched
        int first, n, ncfg, lists[4], pairs;
        if (int rc = tlb_node_plan_shard(nstreams, cfg.data(), G, g, &first, &n, &ncfg, lists, &pairs)) die("illegal configuration", rc);
        std::fprintf(stderr, "nodetick: shard %d on device %d: streams [%d, %d), %d configuration(s), kernel lists psy0/1/2+4/3 = %d/%d/%d/%d\n",
                     g, devices[(size_t)g], first, first + n, ncfg, lists[0], lists[1], lists[2], lists[3]);
    }
    int err = 0;
    tlb_node *nd = tlb_node_create(G, devices.data(), nstreams, cfg.data(), &nc, &err);
    if (!nd) die("tlb_node_create", err);

    std::vector<uint64_t> hash((size_t)G, 1469598103934665603ull);
    std::vector<long> packets((size_t)G, 0), bytes((size_t)G, 0);
    Ctx ctx{nd, &pcm, nframes_in, 0, &hash, &packets, &bytes};
    std::FILE *fo = outpath.empty() ? nullptr : std::fopen(outpath.c_str(), "wb");
    auto tap = [&]() {                                           // -o: the last stream's packets
        if (!fo) return;
        const int s = nstreams - 1;
        for (int u = 0; u < tlb_node_units(nd, s); u++) {
            int len = 0;
            const uint8_t *p = tlb_node_packet(nd, s, u, &len);
            if (!p || !len) continue;
            const uint32_t n = (uint32_t)len;
            const uint8_t le[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
            if (std::fwrite(le, 1, 4, fo) != 4 || std::fwrite(p, 1, n, fo) != n) die("write", 0);
        }
    };

    std::fputs(tlb_node_describe(nd), stderr);                            // which device every shard runs on (name, CUs, XCDs, PCI address, UUID)
    // A GPU that fails takes ITS block off air, not the node (include/toolame_batch.h, FAULT ISOLATION): the call in which a shard
    // breaks returns its code, every other shard has completed the call.  The caller's part: find out which shard, log why, restart it
    // when no tick is in flight -- the reference's "restart the failed input, nothing else stops" (src/odr-audioenc.cpp:875-902) one level up.
    long restarts = 0;
    auto shard_failed = [&](const char *where, int rc) {
        int alive = 0;
        for (int g = 0; g < G; g++) {
            tlb_node_shard_info info;
            if (tlb_node_shard_status(nd, g, &info) == TLB_SHARD_BROKEN) std::fprintf(stderr, "nodetick: %s: shard %d on %s is down (%s)\n", where, g, info.device_name, info.what);
            else alive++;
        }
        if (!alive) die(where, rc);                                       // nobody left: nothing to carry on with
    };
    const auto t0 = std::chrono::steady_clock::now();
    // fill 0, submit 0; then per tick: fill t+1, submit t+1, wait t, ship t
    ctx.tick = 0;
    if (int rc = tlb_node_parallel(nd, fill, &ctx)) die("fill", rc);
    if (int rc = tlb_node_submit(nd)) shard_failed("tlb_node_submit", rc);
    bool in_flight2 = false;
    for (long t = 0; t < ticks; t++) {
        in_flight2 = false;
        if (t + 1 < ticks) {
            ctx.tick = t + 1;
            if (int rc = tlb_node_parallel(nd, fill, &ctx)) die("fill", rc);
            if (int rc = tlb_node_submit(nd)) shard_failed("tlb_node_submit", rc);
            in_flight2 = true;
        }
        if (int rc = tlb_node_wait(nd)) shard_failed("tlb_node_wait", rc);
        if (int rc = tlb_node_parallel(nd, ship, &ctx)) die("ship", rc);
        tap();
        if (!in_flight2)                                                  // no tick in flight: the moment a broken shard may come back
            for (int g = 0; g < G; g++)
                if (tlb_node_shard_status(nd, g, nullptr) == TLB_SHARD_BROKEN && tlb_node_shard_restart(nd, g, -1) == TLB_OK) restarts++;
    }
    if (int rc = tlb_node_finish(nd)) shard_failed("tlb_node_finish", rc); // toolame_finish for every service: the pending last frame
    if (int rc = tlb_node_parallel(nd, ship, &ctx)) die("ship", rc);
    tap();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    std::vector<tlb_node_counter> per((size_t)G);
    tlb_node_counter tot;
    tlb_node_counters(nd, per.data(), &tot);
    uint64_t all = 0;
    long npk = 0, nby = 0;
    for (int g = 0; g < G; g++) {
        std::fprintf(stderr, "nodetick: shard %d (device %d): %ld frames in %ld ticks, busy %.1f ms, %ld packets, %ld bytes\n", g, per[(size_t)g].device,
                     per[(size_t)g].frames, per[(size_t)g].steps, per[(size_t)g].busy_ns / 1e6, packets[(size_t)g], bytes[(size_t)g]);
        all ^= hash[(size_t)g] + 0x9e3779b97f4a7c15ull * (uint64_t)(g + 1);
        npk += packets[(size_t)g]; nby += bytes[(size_t)g];
    }
    // one line for scripts: frames, packets, bytes, a hash of everything shipped (independent of G only per shard -- so print per-stream-order-free totals)
    std::printf("{\"streams\": %d, \"shards\": %d, \"ticks\": %d, \"frames\": %ld, \"packets\": %ld, \"bytes\": %ld, \"seconds\": %.4f, \"frames_per_s\": %.1f, \"realtime_x\": %.2f}\n",
                nstreams, G, ticks, tot.frames, npk, nby, sec, sec > 0 ? tot.frames / sec : 0.0, sec > 0 ? ticks * 0.024 / sec : 0.0);
    (void)all; (void)restarts;
    if (fo) std::fclose(fo);
    tlb_node_destroy(nd);
    return 0;
}
