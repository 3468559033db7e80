// mp2_types.h -- data layout shared by the host runtime and the HIP kernels.
//
// One *stream* = one independent DAB MP2 encoder instance (what the reference keeps in process
// globals, libtoolame-dab/toolame.c:89-118).  Streams are grouped by *config* (sample rate, mode,
// bitrate, psy model); a config carries every table the per-frame path needs, pre-computed on the
// host with the host libm exactly like the reference's init code does.
#pragma once
#include <stdint.h>

#define TL_MAX_FRAME_BYTES 1728      // 384 kbps @ 32 kHz
#define TL_MAX_FRAME_WORDS (TL_MAX_FRAME_BYTES / 4)
#define TL_HIST 480                  // filterbank history (512-32); psy-1/3 need the last 192
#define TL_MAX_XPAD 256              // X-PAD + F-PAD bytes per frame handled on device (the caller accepts padlen 0..255, src/odr-audioenc.cpp:566)

// The part of the common tables that sits on dependent-load chains (dB sums, scalefactor search,
// allocation loop): copied once per workgroup into LDS and shared by its waves.
struct TlBlockShared {
    double dbtable[1002];        // psycho_1.c:170-178; [1000] = -0.0: the addend when the levels are > 99 dB apart
    double scalefactor[64];      // encode_new.c:65-83
    double snr_line[9][16];      // SNR[step_index[line][ba]]             (encode_new.c:16-27,96-100)
    int16_t bits12_line[9][16];  // 12*group*bits of step_index[line][ba] (encode_new.c:1125-1133)
    uint16_t qinfo_line[9][16];  // quantiser class of (line, ba): step_index | bits << 5 | (group == 3) << 10 (encode_new.c:16-100)
    double dct_t[16][2][16];     // matrixing coefficients m[r][2k+par] stored [k][par][r]: one k = 32 consecutive doubles
    uint8_t sfpat[32];           // scalefactor transmission pattern of the class pair 5*c0+c1 (encode_new.c:296-301, ISO Table C.4):
                                 //   source of sf0 | sf1 << 2 | sf2 << 4 (0..2 = sf0..sf2, 3 = min(sf0, sf2)) | scfsi << 6
    double scale_db[64];         // 20*log10(scalefactor[i]*32768) - 10 (psycho_1.c:575, psycho_3.c:180): the same for every configuration, so the
                                 //   SMR line reads it from the workgroup's LDS copy instead of the configuration record in HBM (TlConfig::scale_db)
};

// Small tables of the packing stage that sit on dependent-load chains (quantiser class constants, CRC powers): the encode
// kernel of the split path keeps a copy in LDS, the fused kernels read them where they are (TlTables::pack).
struct TlPackTables {
    double qa[18], qb[18];
    double steps2n_f[18];        // (double)steps2n[q]
    int32_t steps[18];
    int32_t steps2n[18];
    uint16_t crc_xpow[512];      // x^e mod (x^16+x^15+x^2+1), e = 0..511: lets lanes fold CRC-16 chunks in parallel (crc.c:43-56)
    uint8_t crc8_xpow[320];      // x^e mod (x^8+x^4+x^3+x^2+1), the ScF-CRC polynomial (crc.c:99-113)
};

// Tables common to every config.  (ref: enwindow.h, subband.c:125-137, psycho_1.c:170-178,225-233,
// fft.c:38-73,1139-1149, encode_new.c:16-100,448-462)
struct TlTables {
    double enwindow[512];
    double enwindow_s[512];      // enwindow / 32768: the filterbank scales the coefficient instead of the sample (exact, see mp2_wave.h K1)
    double dct[16][32];
    double hann[1024];
    double hann_s[1024];          // hann[i] / 32768 (exact): the model's window with the sample scaling folded in (tl_psy_spectrum)
    double dbtable[1000];
    double fht_tw[166][4];       // (c1,s1,c2,s2) for passes k=2,4,6,8, i=1..kx-1
    double fht_tw_lane[3][128][4];   // the same rows in the order the lanes of passes k=4,6,8 use them: [pass][butterfly g][.]
    uint32_t fht_fg_lane[3][128];    // where butterfly g of the pass is: byte offsets of its points f0 (low half) and g0 (high half) in the transform buffer
    double scalefactor[64];
    double snr[18];
    TlPackTables pack;
    uint8_t bits[18];
    uint8_t group[18];           // 3 = three codewords, 1 = one grouped codeword
    uint8_t step_index[9][16];
    uint8_t nbal_line[9];
    uint8_t pad_[3];
    uint8_t rs_log[256], rs_exp[512];   // GF(2^8), field polynomial 0x11d: log (255 for 0) and antilog (doubled, no modulo)
    uint8_t rs_mlog[207][48];    // log of M[i][j]: parity byte j of the RS(255,207) codeword of the unit chunk e_i (csrc/edi_pft.h)
    uint16_t edi_xpow8[2048];    // x^(8k) mod (x^16+x^12+x^5+1): AF-packet CRC chunks (csrc/edi_af.h; contrib/crc.c:247-255)
    double dblog[1002 + 256];    // what the model phase keeps in LDS: the dB-sum table as in `shared` ([0..1001]) followed by glibc's log table
                                 // (tl_libm.h tlm_log_tab, 128 x {invc, logc} as bit patterns): TL_LOGTAB(db)
    TlBlockShared shared;
};

// Per-config constants.  (ref: toolame.c:120-262, common.c:76-144, encode_new.c:104-125)
struct TlConfig {
    int32_t version, fs_idx, br_idx, kbps, nch, mode0, mode_ext0, tab, sblimit, jsbound0;
    int32_t dab_ext, dab_length, psy, frame_bytes, br_per_ch;   // frame_bytes: slots of a frame WITHOUT the padding slot (availbits.c:49 `whole`)
    double pad_frac;             // availbits.c:50 `frac`: 0 at 48/32/24/16 kHz; at 44.1/22.05 kHz some frames carry one padding slot more
    int32_t p1_ncb, p1_sub, p3_cbands, psy2_tab;
    uint8_t line[32];            // alloc-table line per subband (255 above sblimit)
    uint8_t nbal[32];            // bits of the bit_alloc field per subband
    double scale_db[64];         // 20*log10(scalefactor[i]*32768) - 10   (psycho_1.c:575, psycho_3.c:180)
    // psy model 1 (psycho_1.c:94-168; ISO Table D.1/D.2)
    int16_t p1_cbound[28];
    int16_t p1_line[136];
    double p1_bark[136];
    double p1_hear[136];
    uint8_t p1_map[520];
    uint8_t p1_lineband[520];    // critical band of each FFT line (index into p1_cbound), 255 outside the bands
    double p1_lbark[512], p1_lhear[512];   // p1_bark / p1_hear of the table row each FFT line maps to (one load instead of two dependent ones)
    uint32_t p1_lineinfo[512];   // per FFT line inside the bands: band | lo << 8 | hi << 20 (lo/hi = first line of the band / of the next); 0 outside
    double p1_linerw[512];       // 1 / (hi - lo) of the line's band, correctly rounded (the weight division, psycho_1.c:364-366, as tl_div_by); 0 outside
    int16_t p1_mm_j0[32];        // minimum-mask walk (psycho_1.c:541-559) resolved per subband:
    int16_t p1_mm_n[32];         //   first table row, number of rows (0 => use hear[sub-1])
    // psy model 3 (psycho_3.c:434-512)
    double p3_bark[520];
    double p3_ath[520];
    int16_t p3_cbidx[36];
    int16_t p3_subset[136];
    uint8_t p3_lineband[520];    // critical band of each FFT line (index into p3_cbidx)
    uint32_t p3_lineinfo[520];   // band | lo << 8 | hi << 20 per FFT line, as for psy 1
    int16_t p3_sb_j0[32], p3_sb_n[32];   // rows of p3_subset that fall into each subband (psycho_3.c:415-420)
    // psy model 0 (psycho_0.c:36-50)
    double p0_athmin[32];
};

// psy model 2 tables; they depend on the sample rate only (psycho_2.c:259-420, absthr.h).
#define TL_P2_BAND 48           /* the widest run of non-zero spreading coefficients the device path takes (tl_psy2_band checks every table: 43 at most) */
#define TL_P2_B 8               /* ... requested in batches of eight, two batches ahead of their use (tl_psy2_pass) */
struct TlPsy2Tables {
    double window[1024];         // 0.5*(1-cos(2*PI*(i-0.5)/1024)), psycho_2.c:318-319
    double absthr[513];
    double s_t[64][64];          // spreading function TRANSPOSED: s_t[k][j] = s[j][k]  (coalesced by partition j)
    // The same function as each partition's BAND: row j of s is zero outside a run of at most TL_P2_BAND partitions (43 for model 2's
    // -100 dB cut, 29 for model 4's -60 dB), and a zero coefficient adds +0 to a sum of non-negative terms -- nothing, bit for bit (the
    // reference skips them, psycho_2.c:165).  s_band[q][j] = s[j][band_lo[j] + q], band_lo[j] + TL_P2_BAND <= 64 (tl_psy2_band, mp2_host.cpp).
    double s_band[TL_P2_BAND][64];
    int16_t band_lo[64];
    double tmn[64], bmaxk[64];   // tone-masking-noise, bmax[(int)(cbval+0.5)]
    double den[64];              // rnorm[j]*numlines[j]; 0 => nb[j] = 0  (psycho_2.c:200-204)
    int16_t part_lo[64], part_hi[64];   // FFT lines [lo,hi) of each partition (empty partitions: lo = hi)
    uint8_t partition[520];
    int32_t npart, band_w;       // band_w: the widest run of this table (<= TL_P2_BAND)
};
// psy model 2 prediction state per stream: r and phi of the two previous 576-sample passes (psycho_2.c:300-306)
struct TlPsy2State {
    double r[2][2][513];
    double phi[2][2][513];
};

// What the psy kernel of models 2 and 4 hands to the encode kernel per frame: the SMR per (channel, subband).
struct TlPsyOut { double a[2][32]; };

// Per-stream state that persists across launches.
struct TlStreamState {
    int16_t hist[2][TL_HIST];               // last 480 PCM samples per channel
    uint32_t pending[TL_MAX_FRAME_WORDS];   // previous frame, waiting for its ScF-CRC (toolame.c:527-542)
    int32_t frames_done;
    int32_t pending_len;                    // its length in bytes (frame_bytes, or one more: padding slot)
    double slot_lag;                        // availbits.c:27-33 `slots.lag`, the padding recurrence's state after frames_done frames
};

// Optional per-frame stage taps for parity tests (written only when a tap buffer is given).
struct TlTaps {
    double sb_sample[2][3][12][32];
    double smr[2][32];
    double max_sc[2][32];
    uint32_t subband[2][3][12][32];
    uint8_t scalar_pre[2][3][32];
    uint8_t scalar[2][3][32];
    uint8_t j_scale[3][32];
    uint8_t scfsi[2][32];
    uint8_t bit_alloc[2][32];
    int32_t adb_left, mode, mode_ext, jsbound, crc16;
    uint8_t scfcrc[4];
    int32_t pad_[2];
};

// Launch arguments (plain pointers; device pointers on the GPU, host pointers in the emulation).
struct TlLaunch {
    const TlTables *tables;
    const TlConfig *configs;          // [nconfigs]
    const int32_t *stream_cfg;        // [nstreams] -> config index; NULL: every stream of the batch has configuration 0 (tl_cfg_index: one dependent load less per unit)
    const int32_t *stream_list;       // [nlist] stream ids handled by this launch (the streams of one psy model); NULL: all streams of the batch, position = id
    TlStreamState *state;             // [nstreams]
    const int16_t *pcm;               // [nframes][nstreams][2][1152]
    const uint8_t *xpad;              // [nframes][nstreams][TL_MAX_XPAD] or null
    const int32_t *xpad_len;          // [nframes][nstreams] or null
    uint8_t *out;                     // [nframes][nstreams][out_stride]: slot f holds frame (f-1); slot 0 = pending
    int32_t *out_len;                 // [nframes][nstreams] bytes of the frame in each slot (0: none), or null
    TlTaps *taps;                     // [nframes][nstreams] or null
    long long *stamps;                // [nframes][nstreams][32] cycle stamps (diagnostic builds) or null
    const TlPsy2Tables *psy2_tables;  // [*] indexed by TlConfig::psy2_tab, or null when no stream uses psy 2
    TlPsy2State *psy2_state;          // [nstreams][2] or null: two copies per stream, a launch reads copy psy2_flip and leaves the state in the other
    TlPsyOut *psy_out;                // [nframes][nstreams] psy-2 kernel -> encode kernel (models 2 and 4), or null
    uint8_t *scfcrc;                  // [nframes][nstreams][4] ScF-CRC bytes of each frame (split path: encode kernel -> finish kernel)
    uint32_t *newpend;                // [nstreams][TL_MAX_FRAME_WORDS] last frame of the launch, before it becomes the pending one
    uint8_t *padbits;                 // [nframes][nstreams] padding slot of each frame (tl_slots_stream), or null: no stream of the launch pads
    double *newlag;                   // [nstreams] the slot recurrence's state after the launch (with padbits)
    int32_t *work;                    // unit counters of the persistent kernels: [0] psy-2 kernel, [32 (1 + q)] list q of the eight per-XCD lists of (stream, frame) units
    const int32_t *partner;           // [nstreams] the mono stream of the same configuration a mono stream shares its waves with (tl_encode_pair), -1: none; or null
    const int32_t *chain_list;        // [nchain] psy-2 kernel: stream id | channel << 30 of each (stream, channel) chain of the launch
    int32_t nstreams, nframes, out_stride, nlist;
    int32_t nchain, p2_nwhole, p2_k, p2_plen;    // psy-2 kernel's work list (tl_psy2_unit): whole chains, then runs of p2_plen frames
    int32_t psy2_flip, pad_;
};
