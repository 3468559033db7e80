// tlb_mailbox.h -- one worker thread with a one-slot mailbox: the threading primitive of the node level (csrc/tlb_node.cpp: a Shard IS
// one of these plus the objects it talks to).  post() hands the thread a job, join_job() waits for its return code; ONE poster
// thread per mailbox (the node's caller), one job in the slot at a time.  Header-only and HIP-free so that a CPU test can drive it
// with fake jobs under ThreadSanitizer (tests/test_mailbox_tsan.py, tests/emu/mailbox_tsan.cpp); `on_start` runs once on the new
// thread before the first job (the shard sets its HIP device there).
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

struct TlbMailbox {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, quit = false, done = true;
    int rc = 0;

    void start(std::function<void()> on_start = nullptr)
    {
        th = std::thread([this, on_start] { if (on_start) on_start(); loop(); });
    }
    void loop()
    {
        for (;;) {
            std::function<int()> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return has_job || quit; });
                if (quit && !has_job) return;
                j = std::move(job);
                has_job = false;
            }
            const int r = j();
            {
                std::lock_guard<std::mutex> lk(mu);
                rc = r; done = true;
            }
            cv.notify_all();
        }
    }
    void post(std::function<int()> j)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = std::move(j); has_job = true; done = false;
        }
        cv.notify_all();
    }
    int join_job()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
    // no further job: the thread leaves its loop after the one it may be running
    void stop()
    {
        if (!th.joinable()) return;
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv.notify_all();
        th.join();
    }
};
