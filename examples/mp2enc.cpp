// mp2enc -- a minimal "odr-audioenc -i file -o file" over the batched C-ABI (include/toolame_batch.h): raw interleaved
// s16le PCM in, DAB MP2 frames out, for one stream or for N copies of it in one batch.  Host code only (plain C++, no
// HIP in this file); everything after the file read happens in libtoolame_dab_hip.so:
//
//   file -> [gain, peak, de-interleave: tlb_ingest_host]  (src/odr-audioenc.cpp:1030-1051,1139-1152)
//        -> [encode: tlb_encode_host_len]                 (toolame_encode_frame, libtoolame-dab/toolame.c:267-554)
//        -> whole frames -> file                          (what src/odr-audioenc.cpp:1208-1225 re-frames out of the bursts)
//
// build: g++ -O2 -std=c++17 examples/mp2enc.cpp -Iinclude -Lodr-audioenc_amd -ltoolame_dab_hip -Wl,-rpath,$PWD/odr-audioenc_amd -o mp2enc
// usage: mp2enc in.s16le out.mp2 [-r rate] [-c channels] [-b kbps] [-m s|j|d|m] [-p psy] [-g gain_dB] [-n streams]
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
    std::fprintf(stderr, "mp2enc: %s (code %d)\n", what, code);
    std::exit(1);
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s in.s16le out.mp2 [-r rate] [-c channels] [-b kbps] [-m mode] [-p psy] [-g gain_dB] [-n streams]\n", argv[0]);
        return 2;
    }
    long rate = 48000;
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
        else die("unknown option", 0);
    }
    if (!mode) mode = channels == 1 ? 'm' : 'j';             // odr-audioenc's defaults (src/odr-audioenc.cpp:697-709)
    if (channels != 1 && channels != 2) die("1 or 2 channels", channels);
    if (nstreams < 1) die("streams", nstreams);

    // the whole input, cut to whole frames of 1152 samples per channel
    std::FILE *fi = std::fopen(argv[1], "rb");
    if (!fi) die("cannot open input", 0);
    std::vector<int16_t> in;
    {
        int16_t buf[1 << 15];
        size_t n;
        while ((n = std::fread(buf, sizeof(int16_t), sizeof buf / sizeof buf[0], fi)) > 0) in.insert(in.end(), buf, buf + n);
        std::fclose(fi);
    }
    const size_t per_frame = 1152u * (size_t)channels;
    const int nframes = (int)(in.size() / per_frame);
    if (nframes == 0) die("input shorter than one frame", 0);

    std::vector<tlb_stream_config> cfg((size_t)nstreams, tlb_stream_config{rate, mode, kbps, psy, 0});
    int err = 0;
    tlb_batch *enc = tlb_create(0, nstreams, cfg.data(), &err);
    if (!enc) die("tlb_create", err);
    if (gain_db != 0.0 && (err = tlb_set_gain_db(enc, -1, gain_db)) != TLB_OK) die("tlb_set_gain_db", err);

    const int stride = tlb_out_stride(enc);
    std::FILE *fo = std::fopen(argv[2], "wb");
    if (!fo) die("cannot open output", 0);

    const int chunk = 256;                                    // frames per call
    std::vector<int16_t> inter((size_t)chunk * nstreams * 2304), pcm((size_t)chunk * nstreams * 2304), peaks((size_t)chunk * nstreams * 2);
    std::vector<uint8_t> out((size_t)chunk * nstreams * stride);
    std::vector<int32_t> len((size_t)chunk * nstreams);
    long written = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int f0 = 0; f0 < nframes; f0 += chunk) {
        const int nf = nframes - f0 < chunk ? nframes - f0 : chunk;
        for (int f = 0; f < nf; f++)                         // every stream of the batch gets the same programme
            for (int s = 0; s < nstreams; s++)
                std::memcpy(&inter[((size_t)f * nstreams + s) * 2304], &in[(size_t)(f0 + f) * per_frame], per_frame * sizeof(int16_t));
        if ((err = tlb_ingest_host(enc, inter.data(), nf, pcm.data(), peaks.data())) != TLB_OK) die("tlb_ingest_host", err);
        if ((err = tlb_encode_host_len(enc, pcm.data(), nf, nullptr, nullptr, out.data(), len.data(), nullptr)) != TLB_OK) die("tlb_encode_host_len", err);
        for (int f = 0; f < nf; f++) {                       // stream 0 goes to the file; slot f = the frame before input frame f (length 0: none yet)
            const size_t slot = (size_t)f * nstreams;
            if (len[slot] > 0) { std::fwrite(&out[slot * stride], 1, (size_t)len[slot], fo); written += len[slot]; }
        }
    }
    {   // toolame_finish(): the frame that is still pending
        std::vector<uint8_t> last((size_t)nstreams * stride);
        std::vector<int32_t> llen((size_t)nstreams);
        if ((err = tlb_flush_host_len(enc, last.data(), llen.data())) != TLB_OK) die("tlb_flush_host_len", err);
        if (llen[0] > 0) { std::fwrite(last.data(), 1, (size_t)llen[0], fo); written += llen[0]; }
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fclose(fo);
    std::fprintf(stderr, "mp2enc: %d frames x %d stream(s) in %.3f s = %.0f frames/s (%.0f x real time per stream); %ld bytes written; %s\n",
                 nframes, nstreams, dt, (double)nframes * nstreams / dt, (double)nframes * 1152.0 / (double)rate / dt, written, tlb_version());
    tlb_destroy(enc);
    return 0;
}
