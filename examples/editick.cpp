// editick -- the per-tick loop body of odr-audioenc for N streams over the tick API (include/toolame_batch.h, tlb_tick_*):
// raw interleaved s16le PCM in, the EDI AF packets of stream 0 out (length-prefixed), every tick = one frame of every stream.
// Host code only (plain C++); everything between the two file accesses happens in libtoolame_dab_hip.so:
//
//   file -> pinned host PCM -> [PCIe, gain/peak/de-interleave, encode, EDI AF, PCIe: tlb_tick_run] -> packets -> file
//   (src/odr-audioenc.cpp:1030-1051,1139-1163,1208-1225 and src/Outputs.cpp:194-261 for one stream)
//
// build: g++ -O2 -std=c++17 examples/editick.cpp -Iinclude -Lodr-audioenc_amd -ltoolame_dab_hip -Wl,-rpath,$PWD/odr-audioenc_amd -o editick
// usage: editick in.s16le out.af [-r rate] [-c channels] [-b kbps] [-m s|j|d|m] [-p psy] [-g gain_dB] [-n streams] [-t now_s]
// out.af: for every packet a little-endian uint32 length, then the packet.
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
    std::fprintf(stderr, "editick: %s (code %d)\n", what, code);
    std::exit(1);
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s in.s16le out.af [-r rate] [-c channels] [-b kbps] [-m mode] [-p psy] [-g gain_dB] [-n streams] [-t now_s]\n", argv[0]);
        return 2;
    }
    long rate = 48000;
    long long now_s = 1700000000;
    int channels = 2, kbps = 128, psy = 1, nstreams = 1;
    char mode = 0;
    double gain_db = 0.0;
    for (int i = 3; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        const char *v = argv[i + 1];
        if (k == "-r") rate = std::atol(v);
        else if (k == "-c") channels = std::atoi(v);
        else if (k == "-b") kbps = std::atoi(v);
        else if (k == "-m") mode = v[0];
        else if (k == "-p") psy = std::atoi(v);
        else if (k == "-g") gain_db = std::atof(v);
        else if (k == "-n") nstreams = std::atoi(v);
        else if (k == "-t") now_s = std::atoll(v);
        else die("unknown option", 0);
    }
    if (!mode) mode = channels == 1 ? 'm' : 'j';             // odr-audioenc's defaults (src/odr-audioenc.cpp:697-709)
    if (channels != 1 && channels != 2) die("1 or 2 channels", channels);
    if (nstreams < 1) die("streams", nstreams);

    std::FILE *fi = std::fopen(argv[1], "rb");
    if (!fi) die("cannot open input", 0);
    std::FILE *fo = std::fopen(argv[2], "wb");
    if (!fo) die("cannot open output", 0);

    static const char version[] = "editick example";
    std::vector<tlb_stream_config> cfg((size_t)nstreams, tlb_stream_config{rate, mode, kbps, psy, 0});
    tlb_tick_config tc;
    std::memset(&tc, 0, sizeof tc);
    tc.egress = TLB_TICK_EDI_AF;
    tc.version = version; tc.version_len = (int)std::strlen(version);
    tc.now_s = now_s; tc.delay_ms = 0; tc.tist = 1; tc.tai_utc_offset = 37;
    int err = 0;
    tlb_tick *t = tlb_tick_create(0, nstreams, cfg.data(), &tc, &err);
    if (!t) die("tlb_tick_create", err);
    if (gain_db != 0.0 && tlb_tick_set_gain_db(t, -1, gain_db)) die("gain", 0);

    const size_t per_frame = 1152 * (size_t)channels;        // samples of one frame in the file
    std::vector<int16_t> frame(per_frame);
    long frames = 0, packets = 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto emit = [&]() {
        for (int u = 0; u < tlb_tick_units(t, 0); u++) {
            int len = 0;
            const uint8_t *p = tlb_tick_packet(t, 0, u, &len);
            if (!p || !len) continue;
            const uint32_t n = (uint32_t)len;
            const uint8_t le[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
            if (std::fwrite(le, 1, 4, fo) != 4 || std::fwrite(p, 1, n, fo) != n) die("write", 0);
            packets++;
        }
    };
    while (std::fread(frame.data(), sizeof(int16_t), per_frame, fi) == per_frame) {
        int16_t *in = tlb_tick_pcm(t);                       // pinned [nstreams][2304]; mono streams use the first 1152 values
        for (int s = 0; s < nstreams; s++) std::memcpy(in + (size_t)s * 2304, frame.data(), per_frame * sizeof(int16_t));
        if (int rc = tlb_tick_run(t)) die("tlb_tick_run", rc);
        emit();
        frames++;
    }
    if (frames > 0) {
        if (int rc = tlb_tick_finish(t)) die("tlb_tick_finish", rc);     // the pending last frame (toolame_finish at stream end)
        emit();
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "editick: %ld ticks of %d stream(s), %ld AF packets of stream 0, %.3f s (%.0f frames/s, PCIe and EDI included)\n",
                 frames, nstreams, packets, sec, sec > 0 ? (double)frames * nstreams / sec : 0.0);
    tlb_tick_destroy(t);
    std::fclose(fi);
    std::fclose(fo);
    return 0;
}
