/* mp2_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C11, fp64, re-entrant) of libtoolame-dab's per-frame DAB MP2
 * (MPEG-1/2 Layer II) encode path, i.e. of toolame_encode_frame()
 * (/root/reference/libtoolame-dab/toolame.c:267-554) and everything it calls.
 *
 * Parity status: PINNED.  The reference ships no tests or golden vectors, so the
 * oracle is pinned against the reference itself: oracle/Makefile compiles the reference's own C
 * files (read-only, where they lie) into oracle/_ref/libtoolame_ref.so; tests/golden/make_golden.py
 * runs that library on seeded integer PCM and commits its outputs (bitstreams, per-call return
 * lengths, per-stage taps) under tests/golden/; tests/test_oracle_golden.py checks this restatement
 * against those fixtures byte-for-byte / bit-for-bit.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.  The
 * product path (odr-audioenc_amd/csrc) never includes, links or calls anything under oracle/.
 */
#ifndef MP2_ORACLE_H
#define MP2_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mp2o_enc mp2o_enc;

/* Stage taps of the most recent frame (layout mirrors the reference's statics, toolame.c:96-115). */
typedef struct {
    double sb_sample[2][3][12][32];   /* filterbank output          (subband.c:201)        */
    double j_sample[3][12][32];       /* joint-stereo mid signal    (encode_new.c:237)     */
    unsigned scalar_pre[2][3][32];    /* scalefactor idx before sf_transmission_pattern    */
    unsigned scalar[2][3][32];        /* ... after (what is transmitted)                   */
    unsigned j_scale[3][32];
    double max_sc[2][32];             /* find_sf_max                (encode_new.c:260)     */
    double smr[2][32];                /* psy model output                                   */
    unsigned scfsi[2][32];
    unsigned bit_alloc[2][32];
    unsigned subband[2][3][12][32];   /* quantised samples          (encode_new.c:479)     */
    int adb_left;                     /* stuffing bits              (toolame.c:510)        */
    int mode, mode_ext, jsbound;      /* per-frame header state in joint stereo            */
    unsigned crc16;
    unsigned char scfcrc[4];          /* ScF-CRC bytes in transmission order (i = dab_ext-1..0) */
} mp2o_taps;

/* Configure like odr-audioenc does (src/odr-audioenc.cpp:687-722): init, samplerate, psy, mode,
 * bitrate, pad.  mode in {'s','d','j','m'}; psy in {0,1,2,3} (the ABI-visible range).  Returns NULL for an illegal configuration. */
mp2o_enc *mp2o_create(long samplerate, char mode, int bitrate_kbps, int psy, int pad_len);
void mp2o_destroy(mp2o_enc *e);

int mp2o_frame_bytes(const mp2o_enc *e);   /* lg_frame of the next frame (constant w/o padding) */
int mp2o_nch(const mp2o_enc *e);
int mp2o_sblimit(const mp2o_enc *e);
int mp2o_tablenum(const mp2o_enc *e);
int mp2o_dab_extension(const mp2o_enc *e);

/* Encode one frame with toolame_encode_frame()'s exact output semantics: bytes are appended to
 * the emulated 4096-byte bit buffer and handed out in bursts (bitstream.c:46-71).  Returns the
 * number of bytes written to out (0 is normal). */
int mp2o_encode_frame(mp2o_enc *e, const short pcm[2][1152], const unsigned char *xpad,
                      size_t xpad_len, unsigned char *out, size_t out_size);
/* toolame_finish() (toolame.c:155, bitstream.c:87): flush all whole bytes. */
int mp2o_finish(mp2o_enc *e, unsigned char *out, size_t out_size);

const mp2o_taps *mp2o_get_taps(const mp2o_enc *e);

/* Stage entry points for isolated parity tests. */
void mp2o_filterbank_block(mp2o_enc *e, int ch, const short pcm32[32], double s[32]);
void mp2o_fht1024(double *x);

/* Init-time table taps (enwindow, scalefactor, dct, hann, dbtable, p3_bark, p3_ath, p3_cbidx,
 * p3_subset) as doubles; returns the length or -1. */
int mp2o_get_table(const mp2o_enc *e, const char *name, double *out, int n);
void mp2o_debug_set_p3_power0(double v);

/* Integer-only synthetic PCM (identical in numpy: tests/pcmgen.py).  kind: 0 tones+noise,
 * 1 silence, 2 full-scale square, 3 impulse, 4 full-scale noise, 5 channel-identical tones,
 * 6 low-level (+-1 LSB) noise.  Fills planar pcm[2][1152] for frame index `frame` of stream `seed`. */
void mp2o_gen_pcm(uint32_t seed, int kind, int frame, short pcm[2][1152]);

/* Caller-side glue (gain, positive peak, de-interleave): src/odr-audioenc.cpp:1030-1051,1139-1152. */
unsigned mp2o_silence_ms(unsigned measured_silence_ms, const short peaks[2], int nch, long samplerate);
void mp2o_ingest(const short *in, int nch, double gain_db, short out[2][1152], short peaks[2]);

/* cpu_baseline helper: encode `nframes` frames of stream `seed` (kind 0) and return the number of
 * output bytes (all frames + finish); used only for timing. */
long mp2o_bench_stream(long samplerate, char mode, int kbps, int psy, uint32_t seed, int nframes);

#ifdef __cplusplus
}
#endif
#endif
