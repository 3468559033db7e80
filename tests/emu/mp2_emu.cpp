// mp2_emu.cpp -- TEST-ONLY host emulation of the wave-per-stream HIP encoder.
//
// Compiles odr-audioenc_amd/csrc/mp2_wave.h with -DTL_EMULATE: every TL_LANES_BEGIN/END region
// becomes a loop over 64 lanes, so the CPU test-suite executes the device algorithm (same source,
// same arithmetic, same order) without a GPU.  This library is NOT part of the product: the C-ABI
// library (libtoolame_dab_hip.so) never contains or loads it.
#define TL_EMULATE 1
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../odr-audioenc_amd/csrc/mp2_host.h"
#include "../../odr-audioenc_amd/csrc/mp2_wave.h"
#include "../../odr-audioenc_amd/csrc/edi_af.h"
#include "../../odr-audioenc_amd/csrc/edi_pft.h"

struct Emu {
    TlTables tables;
    std::vector<TlConfig> configs;
    std::vector<int32_t> stream_cfg;
    std::vector<TlStreamState> state;
    std::vector<TlPsy2Tables> psy2_tables;
    std::vector<TlPsy2State> psy2_state;
    int psy2_flip = 0;
    long pair_units = 0;                       // (stream pair, frame) units run by tl_encode_pair so far: the pairing tests read it
};

extern "C" {
void *emu_create(int nstreams, const long *fs, const char *mode, const int *kbps, const int *psy, const int *pad, int *err)
{
    Emu *e = new Emu;
    tl_build_tables(&e->tables);
    e->stream_cfg.resize(nstreams);
    e->state.resize(nstreams);
    memset(e->state.data(), 0, sizeof(TlStreamState) * nstreams);
    for (int s = 0; s < nstreams; s++) {
        // one TlConfig per DISTINCT configuration, as the device batch keeps them (toolame_hip.hip tlb_create): mono streams pair up by config index
        int ci = -1;
        for (int t = 0; t < s && ci < 0; t++)
            if (fs[t] == fs[s] && mode[t] == mode[s] && kbps[t] == kbps[s] && psy[t] == psy[s] && pad[t] == pad[s]) ci = e->stream_cfg[t];
        if (ci < 0) {
            TlConfig c;
            int rc = tl_build_config(&c, fs[s], mode[s], kbps[s], psy[s], pad[s]);
            if (rc) { if (err) *err = rc; delete e; return nullptr; }
            ci = (int)e->configs.size();
            e->configs.push_back(c);
        }
        e->stream_cfg[s] = ci;
    }
    bool any2 = false;
    for (auto &c : e->configs) any2 |= c.psy == 2 || c.psy == 4;
    if (any2) {
        const long rates[TL_PSY2_SLOTS] = {48000, 32000, 24000, 16000, 44100, 22050};
        e->psy2_tables.resize(2 * TL_PSY2_SLOTS);
        for (int i = 0; i < TL_PSY2_SLOTS; i++) {
            tl_build_psy2_tables(&e->psy2_tables[tl_psy2_slot(rates[i])], rates[i]);
            tl_build_psy4_tables(&e->psy2_tables[TL_PSY2_SLOTS + tl_psy2_slot(rates[i])], rates[i]);
        }
        e->psy2_state.resize(2 * (size_t)nstreams);                 // two copies per stream (tl_psy2_chain)
        memset(e->psy2_state.data(), 0, sizeof(TlPsy2State) * 2 * (size_t)nstreams);
    }
    if (err) *err = 0;
    return e;
}
void emu_destroy(void *h) { delete (Emu *)h; }
long emu_pair_units(void *h) { return ((Emu *)h)->pair_units; }
int emu_frame_bytes(void *h, int s) { Emu *e = (Emu *)h; return e->configs[e->stream_cfg[s]].frame_bytes; }
int emu_max_frame_bytes(void *h, int s) { Emu *e = (Emu *)h; const TlConfig &c = e->configs[e->stream_cfg[s]]; return c.frame_bytes + (c.pad_frac != 0 ? 1 : 0); }

int emu_encode_len(void *h, const int16_t *pcm, int nframes, const uint8_t *xpad, const int32_t *xpad_len, uint8_t *out,
                   int out_stride, TlTaps *taps, int32_t *out_len);
int emu_encode(void *h, const int16_t *pcm, int nframes, const uint8_t *xpad, const int32_t *xpad_len, uint8_t *out,
               int out_stride, TlTaps *taps)
{
    return emu_encode_len(h, pcm, nframes, xpad, xpad_len, out, out_stride, taps, nullptr);
}
int emu_encode_len(void *h, const int16_t *pcm, int nframes, const uint8_t *xpad, const int32_t *xpad_len, uint8_t *out,
                   int out_stride, TlTaps *taps, int32_t *out_len)
{
    Emu *e = (Emu *)h;
    TlLaunch A;
    memset(&A, 0, sizeof A);
    A.tables = &e->tables; A.configs = e->configs.data(); A.stream_cfg = e->configs.size() == 1 ? nullptr : e->stream_cfg.data();      // as the device path: one configuration, no table
    A.state = e->state.data(); A.pcm = pcm; A.xpad = xpad; A.xpad_len = xpad_len; A.out = out; A.out_len = out_len; A.taps = taps;
    A.psy2_tables = e->psy2_tables.empty() ? nullptr : e->psy2_tables.data();
    A.psy2_state = e->psy2_state.empty() ? nullptr : e->psy2_state.data();
    A.nstreams = (int)e->state.size(); A.nframes = nframes; A.out_stride = out_stride;
    static thread_local TlMainLds wm;
    static thread_local TlFrameLds wf;
    static thread_local TlPsy2Lds wq;
    // As on the device: the psy-2 kernel's units (models 2 and 4: a chain per channel, second channels first), the (stream, frame)
    // units (any order -- here frames descending within streams ascending, to show that nothing is carried from frame to
    // frame; models 1 and 3: psy model, then encoder, in one LDS block), the finish pass per stream
    std::vector<TlPsyOut> psy_out((size_t)nframes * (size_t)A.nstreams);
    std::vector<uint8_t> scfcrc((size_t)nframes * (size_t)A.nstreams * 4);
    std::vector<uint32_t> newpend((size_t)A.nstreams * TL_MAX_FRAME_WORDS);
    A.psy_out = psy_out.data(); A.scfcrc = scfcrc.data(); A.newpend = newpend.data();
    // 44.1 / 22.05 kHz: the slot recurrence first (sequential per stream), then the units read its padding bits
    std::vector<uint8_t> padbits((size_t)nframes * (size_t)A.nstreams);
    std::vector<double> newlag((size_t)A.nstreams);
    bool pads = false;
    for (int s = 0; s < A.nstreams; s++) pads |= e->configs[e->stream_cfg[s]].pad_frac != 0;
    if (pads) { A.padbits = padbits.data(); A.newlag = newlag.data(); }
    auto model = [&](int s) { return e->configs[e->stream_cfg[s]].psy; };
    {   // the psy-2 kernel's work list as on the device, planned for a machine of `slots` waves (EMU_PSY2_SLOTS, default 3: the
        // last round is cut into runs in nearly every shape) and run in DESCENDING unit order: runs of a chain are independent
        std::vector<int32_t> chains;
        for (int ch = 0; ch < 2; ch++)
            for (int s = 0; s < A.nstreams; s++)
                if ((model(s) == 2 || model(s) == 4) && ch < e->configs[e->stream_cfg[s]].nch) chains.push_back(s | (ch << 30));
        if (!chains.empty()) {
            const char *env = getenv("EMU_PSY2_SLOTS");
            A.chain_list = chains.data(); A.nchain = (int)chains.size(); A.psy2_flip = e->psy2_flip;
            const int nunits = tl_psy2_plan(A.nchain, nframes, env ? atoi(env) : 3, &A.p2_nwhole, &A.p2_k, &A.p2_plen);
            for (int u = nunits - 1; u >= 0; u--) {
                int c, f0, f1;
                if (!tl_psy2_unit(A, u, c, f0, f1)) continue;
                tl_psy2_chain(wq, A, chains[c] & 0x3fffffff, chains[c] >> 30, f0, f1, tlm_sincostab);
            }
            e->psy2_flip ^= 1;
        }
    }
    if (pads) for (int s = 0; s < A.nstreams; s++) tl_slots_stream(A, s);
    // mono streams of one configuration share waves in pairs, as the device path pairs them (toolame_hip.hip batch_build_lists)
    std::vector<int32_t> partner((size_t)A.nstreams, -1);
    {
        std::vector<int> open(e->configs.size(), -1);
        for (int s = 0; s < A.nstreams; s++) {
            const int ci = e->stream_cfg[s];
            if (e->configs[ci].nch != 1) continue;
            if (open[ci] < 0) open[ci] = s;
            else { partner[s] = open[ci]; partner[open[ci]] = s; open[ci] = -1; }
        }
        if (!getenv("EMU_NO_PAIRS")) A.partner = partner.data();
    }
    for (int s = 0; s < A.nstreams; s++)
        for (int f = nframes - 1; f >= 0; f--) {
            int s2;
            if (!tl_unit_partner(A, s, s2)) continue;
            if (s2 >= 0) {
                e->pair_units++;
                if (model(s) == 0) tl_main_pair<0>(wm, &e->tables.shared, e->tables.enwindow_s, &e->tables.pack, A, s, s2, f);
                else if (model(s) == 2 || model(s) == 4) tl_main_pair<2>(wm, &e->tables.shared, e->tables.enwindow_s, &e->tables.pack, A, s, s2, f);
                else if (model(s) == 1) tl_frame_unit<1>(wf, e->tables.dblog, &e->tables.shared, e->tables.enwindow_s, &e->tables.pack, A, &A, s, f, s2);
                else tl_frame_unit<3>(wf, e->tables.dblog, &e->tables.shared, e->tables.enwindow_s, &e->tables.pack, A, &A, s, f, s2);
                continue;
            }
            if (model(s) == 0) tl_main_unit<0>(wm, &e->tables.shared, e->tables.enwindow_s, &e->tables.pack, A, s, f);
            else if (model(s) == 2 || model(s) == 4) tl_main_unit<2>(wm, &e->tables.shared, e->tables.enwindow_s, &e->tables.pack, A, s, f);
            else if (model(s) == 1) tl_frame_unit<1>(wf, e->tables.dblog, &e->tables.shared, e->tables.enwindow_s, &e->tables.pack, A, &A, s, f);
            else tl_frame_unit<3>(wf, e->tables.dblog, &e->tables.shared, e->tables.enwindow_s, &e->tables.pack, A, &A, s, f);
        }
    for (int s = 0; s < A.nstreams; s++) tl_finish_stream(A, s);
    return 0;
}
// The PRODUCT's host-built psy 2 / psy 4 tables (csrc/mp2_host.cpp tl_build_psy2_tables / tl_build_psy4_tables, what the device path uploads)
// by field name, as doubles: window[1024], absthr[513], s[64*64] (s[j][k], i.e. s_t transposed back), tmn / bmaxk / den / part_lo /
// part_hi [64], partition[513], npart[1].  tests/test_oracle_golden.py compares them with the reference's memory.
int emu_psy2_table(long samplerate, int psy, const char *name, double *out, int n)
{
    static TlPsy2Tables P;
    if (psy == 4) tl_build_psy4_tables(&P, samplerate); else tl_build_psy2_tables(&P, samplerate);
    int len = 0;
    std::vector<double> v;
    std::string f(name);
    if (f == "window") v.assign(P.window, P.window + 1024);
    else if (f == "absthr") v.assign(P.absthr, P.absthr + 513);
    else if (f == "s") { v.resize(64 * 64); for (int j = 0; j < 64; j++) for (int k = 0; k < 64; k++) v[(size_t)j * 64 + k] = P.s_t[k][j]; }
    else if (f == "tmn") v.assign(P.tmn, P.tmn + 64);
    else if (f == "bmaxk") v.assign(P.bmaxk, P.bmaxk + 64);
    else if (f == "den") v.assign(P.den, P.den + 64);
    else if (f == "part_lo") v.assign(P.part_lo, P.part_lo + 64);
    else if (f == "part_hi") v.assign(P.part_hi, P.part_hi + 64);
    else if (f == "partition") v.assign(P.partition, P.partition + 513);
    else if (f == "npart") v.assign(1, (double)P.npart);
    else if (f == "band_w") v.assign(1, (double)P.band_w);
    else if (f == "band_lo") v.assign(P.band_lo, P.band_lo + 64);
    else if (f == "s_band") { v.resize(TL_P2_BAND * 64); for (int q = 0; q < TL_P2_BAND; q++) for (int j = 0; j < 64; j++) v[(size_t)j * TL_P2_BAND + q] = P.s_band[q][j]; }   // [j][q]
    else return -1;
    len = (int)v.size();
    if (n < len) return -1;
    memcpy(out, v.data(), sizeof(double) * v.size());
    return len;
}
// The psy-2 kernel's work list for a launch shape: units[nunits][3] = (chain, first frame, end frame); returns nunits (<= cap).
int emu_psy2_units(int nchain, int nframes, int slots, int32_t *units, int cap)
{
    TlLaunch A;
    memset(&A, 0, sizeof A);
    A.nchain = nchain; A.nframes = nframes;
    const int n = tl_psy2_plan(nchain, nframes, slots, &A.p2_nwhole, &A.p2_k, &A.p2_plen);
    if (n != A.p2_nwhole + (nchain - A.p2_nwhole) * A.p2_k) return -1;
    int m = 0;
    for (int u = 0; u < n; u++) {
        int c, f0, f1;
        if (!tl_psy2_unit(A, u, c, f0, f1)) { f0 = f1 = 0; }
        if (m < cap) { units[3 * m] = c; units[3 * m + 1] = f0; units[3 * m + 2] = f1; }
        m++;
    }
    return m;
}
int emu_pending(void *h, int s, uint8_t *out)
{
    Emu *e = (Emu *)h;
    if (e->state[s].frames_done == 0) return 0;
    int n = e->state[s].pending_len;
    for (int i = 0; i < n; i++) out[i] = (uint8_t)(e->state[s].pending[i >> 2] >> (24 - 8 * (i & 3)));
    return n;
}
// EDI AF packets (csrc/edi_af.h) of nframes frames of nstreams streams, emulated wave per stream.
// frames [nframes][nstreams][out_stride], levels [nframes][nstreams][2] or null, state [nstreams] (advanced),
// unit_bytes [nstreams] (3 * kbps; divides frame_bytes), pkts [nframes * max_upf][nstreams][pkt_stride], pkt_len [nframes * max_upf][nstreams]
int emu_edi_af(const uint8_t *frames, const int16_t *levels, int nframes, int nstreams, int out_stride, const int32_t *frame_bytes,
               TlEdiState *state, const uint8_t *version, int version_len, uint8_t *pkts, int pkt_stride, int32_t *pkt_len,
               const int32_t *unit_bytes, int max_upf)
{
    static TlTables T;
    static bool built = false;
    if (!built) { tl_build_tables(&T); built = true; }
    TlEdiArgs A;
    std::vector<TlEdiState> next((size_t)nstreams);
    A.frames = frames; A.levels = levels; A.state = state; A.state_out = next.data(); A.version = version; A.xpow8 = T.edi_xpow8; A.frame_bytes = frame_bytes;
    A.pkts = pkts; A.pkt_len = pkt_len; A.nstreams = nstreams; A.nframes = nframes; A.out_stride = out_stride;
    A.pkt_stride = pkt_stride; A.version_len = version_len; A.unit_bytes = unit_bytes; A.max_upf = max_upf; A.frame_len = nullptr;
    for (int v = 0; v < nframes * max_upf; v++)
        for (int s = 0; s < nstreams; s++) tl_edi_af_packet(A, s, v);
    memcpy(state, next.data(), sizeof(TlEdiState) * (size_t)nstreams);
    return 0;
}
int emu_sizeof_edi_state(void) { return (int)sizeof(TlEdiState); }
// EDI PFT fragments (csrc/edi_pft.h) of nframes AF packets of nstreams streams, emulated wave per packet.
int emu_edi_pft(const uint8_t *af, const int32_t *af_len, int nframes, int nstreams, int af_stride, int fec, int chunk_len, int transport,
                int addr_source, int dest_port, uint16_t *pseq, uint8_t *frags, int32_t *frag_len, int32_t *nfrag, int max_frags, int frag_stride)
{
    static TlTables T;
    static bool built = false;
    if (!built) { tl_build_tables(&T); built = true; }
    std::vector<uint16_t> next((size_t)nstreams);
    TlPftArgs A;
    A.af = af; A.af_len = af_len; A.pseq = pseq; A.pseq_out = next.data();
    A.frags = frags; A.frag_len = frag_len; A.nfrag = nfrag;
    A.nstreams = nstreams; A.nframes = nframes; A.af_stride = af_stride; A.max_frags = max_frags; A.frag_stride = frag_stride;
    A.fec = fec; A.chunk_len = chunk_len; A.transport = transport; A.addr_source = addr_source; A.dest_port = dest_port;
    static TlPftScratch W;
    const TlPftTables R = {T.rs_log, T.rs_exp, &T.rs_mlog[0][0]};
    for (int f = 0; f < nframes; f++)
        for (int s = 0; s < nstreams; s++) tl_edi_pft_packet(A, R, s, f, W);
    memcpy(pseq, next.data(), sizeof(uint16_t) * (size_t)nstreams);
    return 0;
}
// The transform's dealing table of the PRODUCT's host code (csrc/mp2_host.cpp tl_build_tables, TlTables::fht_fg_lane): out[3][128], low half = byte offset of a
// slot's point f0, high half = of g0, 0xffffffff = idle slot.  tests/test_oracle_golden.py checks coverage and the LDS-bank property it is built for.
int emu_fht_dealing(uint32_t *out)
{
    static TlTables T; tl_build_tables(&T);
    memcpy(out, T.fht_fg_lane, sizeof T.fht_fg_lane);
    return (int)(sizeof T.fht_fg_lane / sizeof(uint32_t));
}
int emu_sizeof_taps(void) { return (int)sizeof(TlTaps); }
int emu_sizeof_lds(void) { return (int)sizeof(TlMainLds); }
double emu_log10(double x) { return tlm_log10(x); }
double emu_log10_pn(double x) { return tlm_log10_pn(x, tlm_log_tab); }
double emu_pow10(double x) { return tlm_pow10_sl(x); }
// the allocation code counts instead of searching: needs every allocation line's SNR column to be non-decreasing
int emu_snr_monotone(void)
{
    static TlTables T; tl_build_tables(&T);
    for (int l = 0; l < 9; l++) {
        const int maxa = (1 << T.nbal_line[l]) - 1;
        for (int b = 0; b + 1 < maxa; b++) if (!(T.shared.snr_line[l][b] <= T.shared.snr_line[l][b + 1])) return 0;
    }
    return 1;
}
// tl_sf_index (three reads) against the reference's binary search; returns the number of disagreements
long emu_sf_index_check(const double *v, long n)
{
    static TlTables T; tl_build_tables(&T);
    long bad = 0;
    for (long i = 0; i < n; i++) if (tl_sf_index(T.shared.scalefactor, v[i]) != tl_sf_index_ref(T.shared.scalefactor, v[i])) bad++;
    return bad;
}
// tl_put_bits48 against bit-by-bit writing: n consecutive fields (lengths 1..48) from bit `start`; returns mismatching words
long emu_put_bits48_check(const uint64_t *val, const int32_t *len, long n, int start)
{
    static uint32_t a[TL_MAX_FRAME_WORDS + 2], b[TL_MAX_FRAME_WORDS + 2];
    memset(a, 0, sizeof a); memset(b, 0, sizeof b);
    int pos = start;
    for (long i = 0; i < n && pos + len[i] <= TL_MAX_FRAME_BYTES * 8; i++) {
        const uint64_t v = val[i] & ((len[i] == 64) ? ~0ull : ((1ull << len[i]) - 1ull));
        tl_put_bits48(a, pos, v, len[i]);
        for (int k = 0; k < len[i]; k++)                                  // MSB first
            if ((v >> (len[i] - 1 - k)) & 1ull) b[(pos + k) >> 5] |= 1u << (31 - ((pos + k) & 31));
        pos += len[i];
    }
    long bad = 0;
    for (int w = 0; w < TL_MAX_FRAME_WORDS + 2; w++) bad += a[w] != b[w];
    return bad;
}
void emu_scalefactors(double *out) { static TlTables T; tl_build_tables(&T); for (int i = 0; i < 64; i++) out[i] = T.scalefactor[i]; }
// tl_div_by against the division it replaces; returns the number of mismatching quotients among n (s[i], d[i]) pairs
long emu_div_by_check(const double *s, const double *d, long n)
{
    long bad = 0;
    for (long i = 0; i < n; i++) {
        const double q = tl_div_by(s[i], d[i], 1.0 / d[i]), t = s[i] / d[i];
        uint64_t a, b; memcpy(&a, &q, 8); memcpy(&b, &t, 8);
        if (a != b && !(t == 0.0 && q == 0.0)) bad++;
    }
    return bad;
}
}

extern "C" long emu_near1_full_flushes(void) { return tl_emu_near1_full; }      // full passes of the power spectrum's deferral list so far (mp2_wave.h)
#ifdef TL_DEBUG_DUMP
// diagnostic builds only (-DTL_DEBUG_DUMP): rounds of the tone walks since the process started (tl_psy1_front / tl_psy3_front)
extern "C" long emu_walk_rounds(void) { return tl_dbg_rounds; }
extern "C" void emu_walk_stats(long *out) { out[0] = tl_dbg_rounds; out[1] = tl_dbg_tones; out[2] = tl_dbg_deadheads; out[3] = tl_dbg_fronts; out[4] = tl_dbg_cands; }
#endif
