// tlb_plan.h -- the ONE statement of what a block of streams becomes: which streams share a configuration record and which mono
// streams share waves in pairs.  Used by tlb_create / tlb_stream_reconfigure (csrc/tlb_batch.cpp) and by the node level's planner
// (tlb_node_plan_shard, csrc/tlb_node.cpp), so that the plan a caller is shown cannot drift from what the batch does.  Host C++ only.
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/toolame_batch.h"
#include "mp2_host.h"

// the knobs of toolame.h:13-48 that make a configuration record (the sixth, the PAD length of a frame, is per frame)
static inline bool tlb_same_config(const tlb_stream_config &a, const tlb_stream_config &b)
{
    return a.samplerate == b.samplerate && a.mode == b.mode && a.bitrate == b.bitrate && a.psy_model == b.psy_model && a.pad_len == b.pad_len;
}
static inline int tlb_find_config(const std::vector<tlb_stream_config> &uniq, const tlb_stream_config &c)
{
    for (size_t u = 0; u < uniq.size(); u++) if (tlb_same_config(uniq[u], c)) return (int)u;
    return -1;
}
// streams -> records: streams with the same knobs share one (keeps the tables cache resident); appends to uniq / configs.
// Returns 0 or the TLB_ERR_* of the first illegal configuration (nothing is appended for it).
static inline int tlb_plan_configs(int n, const tlb_stream_config *cfgs, std::vector<tlb_stream_config> &uniq, std::vector<TlConfig> &configs,
                                   std::vector<int32_t> &stream_cfg)
{
    stream_cfg.resize((size_t)n);
    for (int s = 0; s < n; s++) {
        int found = tlb_find_config(uniq, cfgs[s]);
        if (found < 0) {
            TlConfig c;
            if (int rc = tl_build_config(&c, cfgs[s].samplerate, cfgs[s].mode, cfgs[s].bitrate, cfgs[s].psy_model, cfgs[s].pad_len)) return rc;
            uniq.push_back(cfgs[s]); configs.push_back(c);
            found = (int)uniq.size() - 1;
        }
        stream_cfg[(size_t)s] = found;
    }
    return 0;
}
// mono streams of the same record (hence the same model and kernel) in pairs: consecutive ones of the stream order; -1 = alone.
// Returns the number of pairs.
static inline int tlb_plan_pairs(const std::vector<TlConfig> &configs, const std::vector<int32_t> &stream_cfg, std::vector<int32_t> &partner)
{
    const size_t n = stream_cfg.size();
    partner.assign(n, -1);
    std::vector<int> open(configs.size(), -1);                       // per record: a mono stream still waiting for a partner
    int pairs = 0;
    for (size_t s = 0; s < n; s++) {
        const int ci = stream_cfg[s];
        if (configs[(size_t)ci].nch != 1) continue;
        if (open[(size_t)ci] < 0) open[(size_t)ci] = (int)s;
        else { partner[s] = open[(size_t)ci]; partner[(size_t)open[(size_t)ci]] = (int32_t)s; open[(size_t)ci] = -1; pairs++; }
    }
    return pairs;
}
// the kernel list a stream's model rides in: model 4 runs the psy-2 kernels on its own tables
static inline int tlb_model_list(int psy) { return psy == 4 ? 2 : psy; }
