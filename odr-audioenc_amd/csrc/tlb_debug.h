/* tlb_debug.h -- TEST BUILDS ONLY.  Fault injection for the error paths no healthy GPU takes: compiled into the library only with
 * -DTLB_FAULT_INJECT (csrc/Makefile target `fault`: odr-audioenc_amd/libtoolame_dab_hip_fi.so, the product's kernel objects + the host
 * files rebuilt with the flag).  The product library has none of this: no symbol, no branch (tests/test_abi_symbols.py checks the list).
 * An armed launch fails exactly as a failing device call inside it would: tlb_launch returns TLB_ERR_HIP after its guard has marked
 * the batch broken. */
#pragma once
#include "../../include/toolame_batch.h"
#ifdef __cplusplus
extern "C" {
#endif
int tlb_debug_fail_next(tlb_batch *b, int nth);                    /* the nth launch of this batch from now fails (1 = the next; 0 disarms) */
int tlb_debug_tick_fail_next(tlb_tick *t, int nth);                /* ... the nth submit of this tick object, in its LAST group: the groups before it have been queued */
int tlb_debug_node_fail_next(tlb_node *nd, int shard, int nth);    /* ... of one shard of a node */
#ifdef __cplusplus
}
#endif
