// mp2_host.h -- host-side construction of the encoder tables (init-time work of the reference:
// toolame_init/toolame_set_* toolame.c:120-262, hdr_to_frps common.c:76, encode_init
// encode_new.c:104, create_dct_matrix subband.c:125, psycho_1 init psycho_1.c:39-56,94-178,225-233,
// psycho_3_init psycho_3.c:434-512, psycho_0 init psycho_0.c:36-50).
//
// These tables are computed ONCE on the host with the host libm -- exactly where and how the
// reference computes them -- and uploaded to HBM; the per-frame device path never calls libm.
#pragma once
#include "mp2_types.h"

// error codes of tl_build_config (non-zero like the reference's setters, toolame.h:9-10)
enum {
    TL_OK = 0,
    TL_ERR_SAMPLERATE = 1,    // SmpFrqIndex (common.c:118-144) rejects it, or it needs padding slots
    TL_ERR_MODE = 2,          // toolame_set_channel_mode (toolame.c:195-197)
    TL_ERR_PSY = 3,           // toolame_set_psy_model (toolame.c:204-207)
    TL_ERR_BITRATE = 4,       // BitrateIndex (common.c:95-116; the reference exit(-1)s)
    TL_ERR_PAD = 5,           // toolame_set_pad (toolame.c:252-255)
};

void tl_build_tables(TlTables *T);
// psy-2 tables for a sample rate (48000/32000/24000/16000); returns the table slot 0..2 used as TlConfig::psy2_tab
#define TL_PSY2_SLOTS 6      // sample rates with their own psy-2 / psy-4 tables: 48, 32, 24, 16, 44.1, 22.05 kHz
int tl_psy2_slot(long samplerate);
void tl_build_psy2_tables(TlPsy2Tables *P, long samplerate);
void tl_build_psy4_tables(TlPsy2Tables *P, long samplerate);   // psy model 4 mapped onto the psy-2 record
// Work list of the psy-2 kernel for a launch of `nchain` (stream, channel) chains of `nframes` frames on `slots` resident waves:
// chains [0, *nwhole) are one unit each, every later chain is cut into *k runs of *plen frames (tl_psy2_unit in mp2_wave.h).
// Returns the number of units.
int tl_psy2_plan(int nchain, int nframes, int slots, int *nwhole, int *k, int *plen);
int tl_build_config(TlConfig *C, long samplerate, char mode, int kbps, int psy, int pad_len);
