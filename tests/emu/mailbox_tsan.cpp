// mailbox_tsan.cpp -- the node level's threading primitive (csrc/tlb_mailbox.h: one worker thread, a one-slot mailbox) driven with fake
// jobs the way csrc/tlb_node.cpp drives its shards, for ThreadSanitizer (sanitizers belong on the CPU build; tests/test_mailbox_tsan.py
// builds this with -fsanitize=thread and expects a clean exit).  What the node does with a mailbox: post one job to every shard and
// join them all ("live" / "all"), post to one shard ("one"), jobs that write the shard's own counters which the poster reads after
// the join, a job that marks its shard broken so that the next round skips it, restart of that shard, teardown with a last job.
#include <stdio.h>

#include <deque>
#include <functional>
#include <vector>

#include "../../odr-audioenc_amd/csrc/tlb_mailbox.h"

struct FakeShard : TlbMailbox {
    int index = 0;
    long steps = 0, frames = 0;                  // written inside jobs (worker thread), read by the poster after join_job()
    std::deque<double> in_flight;
    bool broken = false, started = false;
};

static int all(std::vector<FakeShard *> &sh, const std::function<int(FakeShard &)> &fn)
{
    std::vector<FakeShard *> on;
    for (FakeShard *s : sh) if (!s->broken) { on.push_back(s); s->post([s, &fn] { const int r = fn(*s); if (r) s->broken = true; return r; }); }
    int rc = 0;
    for (FakeShard *s : on) { const int r = s->join_job(); if (r && !rc) rc = r; }
    return rc;
}

int main(int argc, char **argv)
{
    const int nshards = 6, rounds = argc > 1 ? atoi(argv[1]) : 3000;
    std::vector<FakeShard *> sh;
    for (int g = 0; g < nshards; g++) {
        FakeShard *s = new FakeShard;
        s->index = g;
        sh.push_back(s);
        s->start([s] { s->started = true; });
    }
    long total = 0, broke = 0, restarted = 0;
    for (int r = 0; r < rounds; r++) {
        // "submit": every live shard queues a step; shard (r % 7 == 3 ? 2 : none) fails
        const int rc = all(sh, [&](FakeShard &s) { if (s.index == 2 && r % 7 == 3) { s.in_flight.clear(); return 17; } s.in_flight.push_back((double)r); return 0; });
        if (rc) broke++;
        // "wait": retire it, count
        all(sh, [&](FakeShard &s) { if (s.in_flight.empty()) return 0; s.in_flight.pop_front(); s.steps++; s.frames += 100 + s.index; return 0; });
        // the poster reads the shards' counters between jobs (tlb_node_counters)
        for (FakeShard *s : sh) total += s->steps;
        // "one": a life-cycle call on one shard
        FakeShard *o = sh[(size_t)(r % nshards)];
        if (!o->broken) { o->post([o] { o->frames += 1; return 0; }); o->join_job(); }
        // "restart" of a broken shard, on its own thread
        if (sh[2]->broken && r % 7 == 5) { FakeShard *b = sh[2]; b->post([b] { b->in_flight.clear(); return 0; }); if (b->join_job() == 0) { b->broken = false; restarted++; } }
    }
    long steps = 0;
    for (FakeShard *s : sh) {
        s->post([s] { s->in_flight.clear(); return 0; });              // teardown on the thread that made the objects
        s->join_job();
        s->stop();
        if (!s->started) return 2;
        steps += s->steps;
        delete s;
    }
    printf("mailbox ok: %d rounds, %ld steps, broke %ld times, restarted %ld, checksum %ld\n", rounds, steps, broke, restarted, total);
    return broke > 0 && restarted > 0 ? 0 : 1;
}
